"""Makes tests/golden/circular_array.json from the reference's own CircularArray (main.py:21-53).

main.py cannot be imported here (it imports googleapiclient at module level), so the class definition is taken out of
the reference file with `ast` and executed on its own; the fixture is the trace of a scripted sequence of operations
(appends past capacity, extend, random.shuffle between refills -- the way main.py:89-105 uses it).  Run in this
container only:  python tests/golden/gen_loop_golden.py
"""
import ast
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OZ_REFERENCE", "/root/reference")


def script(cls):
    """the operation sequence; returns the list of observed states"""
    trace = []
    rnd = random.Random(7)
    for cap in (1, 5, 8):
        ca = cls(cap)
        nxt = 0
        for rnd_round in range(6):
            k = rnd.randint(1, cap + 3)
            items = list(range(nxt, nxt + k))
            nxt += k
            if rnd_round % 2:
                ca.extend(items)
            else:
                for it in items:
                    ca.append(it)
            trace.append(dict(cap=cap, op="fill", items=items, state=list(ca), length=len(ca), index=ca._index, rep=repr(ca), s=str(ca)))
            random.seed(100 * cap + rnd_round)
            random.shuffle(ca)
            trace.append(dict(cap=cap, op="shuffle", seed=100 * cap + rnd_round, state=list(ca), first=ca[0], last=ca[-1], sl=ca[0:2]))
            ca[0] = -nxt
            trace.append(dict(cap=cap, op="setitem", value=-nxt, state=list(ca)))
    return trace


if __name__ == "__main__":
    src = open(os.path.join(REF, "main.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "CircularArray")
    ns = {}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "reference:main.py:CircularArray", "exec"), ns)
    out = os.path.join(HERE, "circular_array.json")
    with open(out, "w") as f:
        json.dump(script(ns["CircularArray"]), f)
    print(out, os.path.getsize(out), "bytes")
