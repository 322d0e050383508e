cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
cat > /tmp/pl.py <<'PY'
import os, sys, time, traceback
try:
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    import numpy as np
    from othellozero_amd.NNet import NNetWrapper
    prec = sys.argv[1]
    net = NNetWrapper((8, 8), num_channels_1=512, max_batch=1, seed=0, precision=prec)
    own = np.array([0x0000000810000000], dtype=np.uint64); opp = np.array([0x0000001008000000], dtype=np.uint64)
    for _ in range(300): net.predict_batch(own, opp)
    open(f"/tmp/pl_{prec}.done", "w").write("ok")
except BaseException:
    open(f"/tmp/pl_{sys.argv[1]}.done", "w").write(traceback.format_exc())
PY
for p in f32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/pl_$p -o run -- python3 /tmp/pl.py $p > $O/pl_$p.log 2>&1
  cat /tmp/pl_$p.done; echo
  ls -la $O/pl_$p/
  python tools/trace_gaps.py $O/pl_$p k_conv2_lut_xcd 200 2>&1 | tee $O/predict_gaps_$p.txt
  wc -l $O/pl_$p/*kernel_trace.csv
  python3 - $O/pl_$p <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
c = collections.Counter(r["Kernel_Name"].split("(")[0][:50] for r in csv.DictReader(open(f)))
for k, v in c.most_common(25): print(v, k)
PY
  find $O/pl_$p -type f -delete
done
