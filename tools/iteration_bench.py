"""Wall time of ONE iteration of the reference's training loop at main.py's default settings (board 6, 100 episodes x 25
simulations, buffer 76 800, 10 epochs at batch 32, 10 new-vs-old games, 12 + 12 evaluation games against the random agent),
run through othellozero_amd.loop.training on one GPU.  Prints one JSON line with the phase times."""
import json, logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from othellozero_amd import loop
from othellozero_amd.NNet import NNetWrapper

marks = []
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); marks.append((name, time.perf_counter() - t)); return r
    return w
loop.selfplay_batch = timed("episodes (100 games x 25 sims, lock step)", loop.selfplay_batch)
loop.examples_from_records = timed("records -> example tuples", loop.examples_from_records)
loop.self_play_match = timed("new-vs-old arena (10 games)", loop.self_play_match)
loop.evaluate_against_random = timed("evaluation vs random, sequential drop-in agents (12 games)", loop.evaluate_against_random)
loop.evaluate_against_random_batch = timed("evaluation vs random, lock-step arena (12 games)", loop.evaluate_against_random_batch)

batched = "--batched-eval" in sys.argv
n = 6
precision = "f32" if "--f32" in sys.argv else "f16x2"        # inference AND training arithmetic of the wrapper
net = NNetWrapper((n, n), num_channels_1=512, batch_size=32, epochs=10, max_batch=128, precision=precision)
net.train = timed("fit (10 epochs, batch 32)", net.train)
os.chdir(tempfile.mkdtemp())
t0 = time.perf_counter()
hist = loop.training(board_size=n, num_iterations=1, num_episodes=100, num_simulations=25, degree_exploration=1, temperature=1,
                     neural_network=net, e_greedy=0.9, evaluation_interval=1, evaluation_iterations=12, temperature_threshold=25,
                     self_play_training=True, self_play_interval=1, self_play_total_games=10, self_play_threshold=6,
                     checkpoint_filepath="./othelo_model_weights.h5", training_buffer_size=8 * 32 * 100 * 3, seed=1,
                     batched_evaluation=batched)
total = time.perf_counter() - t0
out = {"metric": "seconds_per_training_iteration", "value": total, "settings": "main.py defaults (board 6)", "batched_eval": batched, "precision": precision,
       "phases": {}, "historic": hist}
for name, dt in marks:
    out["phases"][name] = round(out["phases"].get(name, 0.0) + dt, 3)
print(json.dumps(out))
