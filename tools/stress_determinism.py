#!/usr/bin/env python3
"""Developer tool: repeat the same few bf16 training steps (fresh model each time, fixed seeds) and count the runs whose
losses differ from the first one.  usage: stress_determinism.py [runs] [ride 0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import embracenet_amd as ea
from embracenet_amd import optim, training
from helpers import model_batch, model_fill
from oracle.configs import CONFIGS, FixedTrial
DEV = "cuda"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ride = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False

def build(cfg_name, tag):
    hp, F_in = CONFIGS[cfg_name]
    model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="A549", task="active_E_vs_inactive_E", device=DEV, in_features_FFNN=F_in)
    fill = model_fill(tag)
    model = model.double()
    with torch.no_grad():
        for key, t in model.state_dict().items():
            if "running_" in key or "num_batches" in key:
                continue
            t.copy_(torch.from_numpy(fill(key, tuple(t.shape))))
    return model.to(torch.float32).to(DEV).set_rng("host"), F_in

def run():
    model, F_in = build("cfg1", "rd")
    model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=5)
    model.ride_prenets = ride
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    model.train()
    runner = training.StepRunner(model, opt, DEV)
    table = ea.metrics.StepTable(4, DEV)
    for k in range(4):
        a, b, y = model_batch(f"rd/{k}", 96 if k < 3 else 40, F_in, 0.3)
        runner.train_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
    return table.fetch()[0].tolist()

ref = run()
bad = 0
for i in range(runs):
    if i % 3 == 0:   # perturb the allocator / leave garbage behind
        junk = torch.randn(1 << (18 + i % 5), device=DEV); del junk
    l = run()
    if l != ref:
        bad += 1
        print("run", i, "differs:", l, flush=True)
print(f"ride={ride} runs={runs} differing={bad} ref={ref}")
