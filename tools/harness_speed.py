#!/usr/bin/env python3
"""Developer tool: samples/s of the HARNESS loop (training.StepRunner, the loop fit_multimodal runs) on bench.py's
cfg2 workload, eager vs hipGraph-replayed steps, with batches resident on the device or on the host (pinned).
Not part of the product; bench.py stays the contract measurement."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import embracenet_amd as ea
from embracenet_amd import optim, training


def run(graph, where, packed, epochs=4, nb=200):
    wl = bench.WORKLOADS["cfg2"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = ea.EmbraceNetMultimodal(bench.DictTrial(wl["hp"]), cell_line="A549", task="active_E_vs_inactive_E", device=dev,
                                    in_features_FFNN=wl["F"])
    model = training.prepare_model(model, dev, wl["dtype"]).set_rng("philox", seed=2024)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    batches = []
    for k in range(8):                                        # 8 distinct batches cycled nb times per epoch
        x1, x2, y = bench.synth_batch(wl["B"], wl["F"], wl["pos"], dev, 100 + k)
        x2 = ea.functional.pack_onehot(x2) if packed else x2.to(torch.bfloat16)
        x1 = x1.to(torch.bfloat16)
        if where == "host":
            x1, x2, y = (t.cpu().pin_memory() for t in (x1, x2, y))
        batches.append((x1, x2, y))
    runner = training.StepRunner(model, opt, dev, graph=graph)
    table = ea.metrics.StepTable(nb + 2, dev)
    model.train()
    rates = []
    for ep in range(epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(nb):
            x1, x2, y = batches[i % len(batches)]
            runner.train_step(x1, x2, y, table)
        table.fetch()                                         # the one device->host copy of the epoch
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rates.append(nb * wl["B"] / dt)
    return max(rates[1:])


def run_resident(graph, packed, epochs=4, rows=1024 * 128):
    """The split staged in HBM (data.device_loaders, balanced sampler): per step two gather launches, no host traffic."""
    from embracenet_amd import data
    wl = bench.WORKLOADS["cfg2"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = ea.EmbraceNetMultimodal(bench.DictTrial(wl["hp"]), cell_line="A549", task="active_E_vs_inactive_E", device=dev,
                                    in_features_FFNN=wl["F"])
    model = training.prepare_model(model, dev, wl["dtype"]).set_rng("philox", seed=2024)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    x1, x2, y = bench.synth_batch(rows, wl["F"], wl["pos"], dev, 7)
    loaders = data.device_loaders(x1, x2, y, wl["B"], dev, balanced=True, feature_dtype=torch.bfloat16, pack_sequence=packed)
    runner = training.StepRunner(model, opt, dev, graph=graph)
    table = ea.metrics.StepTable(len(loaders["FFNN"]) + 2, dev)
    model.train()
    rates = []
    for ep in range(epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for a, b, t in training._pairs(loaders):
            runner.train_step(a, b, t, table)
            n += len(t)
        table.fetch()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rates.append(n / dt)
    return max(rates[1:])


if __name__ == "__main__":
    out = {}
    for packed in (False, True):
        for graph in (False, True):
            out[f"resident split/{'codes' if packed else 'onehot'}/{'graph' if graph else 'eager'}"] = round(run_resident(graph, packed))
    for where in ("device", "host"):
        for packed in (False, True):
            for graph in (False, True):
                out[f"{where}/{'codes' if packed else 'onehot'}/{'graph' if graph else 'eager'}"] = round(run(graph, where, packed))
    print(json.dumps(out, indent=1))
