#!/usr/bin/env python3
"""Headline benchmark: EmbraceNetMultimodal training step throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full pass of the hot path over one synthetic batch: zero_grad, forward
(pre-nets -> fused docking/ReLU/selection -> post stack), class-weighted cross-entropy, backward, gradient
all-reduce (N > 1), optimizer step -- the semantics of the reference's train step
(BIOINF_tesi/models/utils/training_models_multimodal.py:132-163) without its per-step host syncs.
Workload = BASELINE.json configs[1]: A549 two-modality EmbraceNet (trial-0 pre-nets of the Optuna DB: d0=16,
d1=1856), docking dim c=256, per-GPU batch 1024, bf16 storage with fp32 accumulate.  Weak scaling: every GPU
keeps 1024 rows, the global batch is 1024*N.  Inputs are resident in HBM before the timed region.

One JSON line on stdout (rank 0); see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a torchrun environment: this parent never touches a device; it starts
    N fresh rank processes the way the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same arguments>`) as a CHILD process (no exec), relays rank 0's JSON
    line on stdout and everything else on stderr, and returns the child's exit code (non-zero when any rank failed; no
    retry)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    env["EMB_BENCH_LAUNCHED"] = str(n)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script or os.path.abspath(__file__), *argv]
    print(f"[bench] --gpus {n} without a torchrun environment: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        print(f"[bench] the rank processes did not finish within {timeout} s", file=sys.stderr, flush=True)
        sys.stdout.write(e.stdout if isinstance(e.stdout, str) else (e.stdout or b"").decode())
        return 124
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln                                     # rank 0's result (the last one, should a rank have echoed another)
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        print("[bench] the rank processes exited cleanly but printed no result line", file=sys.stderr, flush=True)
        return 1
    return r.returncode


def _launch_if_needed(argv):
    """Runs before `import torch`: the parent of a self-launched N > 1 run imports nothing that could touch a device."""
    n, force = 1, False
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
        elif a == "--force-collectives":
            force = True
    if n > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and not force:
        sys.exit(launch_ranks(n, argv))


if __name__ == "__main__":
    _launch_if_needed(sys.argv[1:])

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(name="A549 2-modality EmbraceNet, c=256, B=1024/GPU, bf16", F=48, B=1024, dtype="bfloat16", pos=0.1,
                 hp=dict(FFNN_n_layers=3, FFNN_n_units_l0=32, FFNN_dropout_l0=0.0, FFNN_n_units_l1=16, FFNN_dropout_l1=0.0,
                         FFNN_n_units_l2=16, FFNN_dropout_l2=0.0, CNN_n_layers=2, CNN_out_channels_l0=64,
                         CNN_kernel_size_l0=15, CNN_dropout_l0=0, CNN_out_channels_l1=32, CNN_kernel_size_l1=15,
                         CNN_dropout_l1=0, EMBRACENET_embracement_size=256, n_post_layers=0,
                         selection_probabilities_FFNN=0.5784523087676721)),
}
WORKLOADS["cfg1"] = dict(WORKLOADS["cfg2"], name="A549 2-modality EmbraceNet, c=512, B=64, fp64 (reference plumbing case)",
                         B=64, dtype="float64", hp=dict(WORKLOADS["cfg2"]["hp"], EMBRACENET_embracement_size=512))
WORKLOADS["cfg4"] = dict(name="K562 E-vs-P, c=1024, d1=1024, n_post=2, fp32, B=1024/GPU (MFMA-bound docking)", F=429, B=1024,
                         dtype="float32", pos=0.306,
                         hp=dict(FFNN_n_layers=1, FFNN_n_units_l0=64, FFNN_dropout_l0=0.0, CNN_n_layers=4,
                                 CNN_out_channels_l0=32, CNN_kernel_size_l0=5, CNN_dropout_l0=0, CNN_out_channels_l1=32,
                                 CNN_kernel_size_l1=5, CNN_dropout_l1=0, CNN_out_channels_l2=128, CNN_kernel_size_l2=11,
                                 CNN_dropout_l2=0, CNN_out_channels_l3=128, CNN_kernel_size_l3=15, CNN_dropout_l3=0,
                                 EMBRACENET_embracement_size=1024, n_post_layers=2, EMBRACENET_n_units_l0=256,
                                 EMBRACENET_dropout_l0=0.0, EMBRACENET_n_units_l1=128, EMBRACENET_dropout_l1=0.0,
                                 selection_probabilities_FFNN=0.5))

# BASELINE.json configs[2] as SURVEY 8d defines it ("7 cell lines concatenated": F = 562 = the widest line, the nets of cfg1
# with c = 768, fp32, global batch 8192 -> 1024 rows per GPU)
WORKLOADS["cfg3"] = dict(name="7 cell lines (F=562), 2-modality EmbraceNet, c=768, B=1024/GPU, fp32", F=562, B=1024,
                         dtype="float32", pos=0.1, hp=dict(WORKLOADS["cfg2"]["hp"], EMBRACENET_embracement_size=768))
# BASELINE.json configs[4] as SURVEY 8d defines it: GM12878 (F = 152), c = 768, d0 = 32, d1 = 3712, modality dropout (the
# model's own training behaviour, EmbraceNetMultimodal.py:178-182), global batch 4096 -> 512 rows per GPU, graph-captured step
WORKLOADS["cfg5"] = dict(name="GM12878 (F=152) 2-modality EmbraceNet with modality dropout, c=768, d0=32, d1=3712, B=512/GPU, bf16",
                         F=152, B=512, dtype="bfloat16", pos=0.183,
                         hp=dict(FFNN_n_layers=2, FFNN_n_units_l0=64, FFNN_dropout_l0=0.0, FFNN_n_units_l1=32, FFNN_dropout_l1=0.0,
                                 CNN_n_layers=2, CNN_out_channels_l0=64, CNN_kernel_size_l0=11, CNN_dropout_l0=0,
                                 CNN_out_channels_l1=64, CNN_kernel_size_l1=11, CNN_dropout_l1=0,
                                 EMBRACENET_embracement_size=768, n_post_layers=0, selection_probabilities_FFNN=0.5))
# the cfg2 network at a 4x larger per-GPU batch (what a single GPU does with the whole 4096-row global batch of configs[4])
WORKLOADS["cfg2_b4096"] = dict(WORKLOADS["cfg2"], name="A549 2-modality EmbraceNet, c=256, B=4096/GPU, bf16 (cfg2 network, 4x batch)",
                               B=4096)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable
MFMA_PEAK_TFLOPS = {"bfloat16": 2500.0, "float32": 157.3, "float64": 78.6}


class DictTrial:
    def __init__(self, p):
        self.p = p

    def suggest_int(self, n, lo, hi):
        return self.p[n]

    def suggest_categorical(self, n, ch):
        return self.p[n]

    def suggest_float(self, n, lo, hi):
        return self.p[n]


def synth_batch(B, F, pos, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    x1 = torch.rand(B, F, device=device, generator=g)
    base = torch.randint(0, 4, (B, 256), device=device, generator=g)
    x2 = torch.nn.functional.one_hot(base, 4).permute(0, 2, 1).contiguous().float()
    y = (torch.rand(B, device=device, generator=g) < pos).long()
    return x1, x2, y


def event_time_us(fn, iters, warm=5):
    """Average device time of one fn() call: `iters` calls are captured into ONE hipGraph and the replay is bracketed by HIP
    events on the launch stream -- back-to-back launches with no host launch overhead in between (launched one by one
    from Python the ~20 us calls are host-bound on a slow box)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / iters


def kernel_roofline(ea, wl, device):
    """Isolated timing of the two hand-written GEMM-class kernels of the step on this workload's shapes, with HIP
    events on the launch stream; algorithmic bytes/flops per launch as defined in DESIGN.md."""
    F = ea.functional
    hp, B = wl["hp"], wl["B"]
    T = {"bfloat16": torch.bfloat16, "float32": torch.float32, "float64": torch.float64}[wl["dtype"]]
    P = torch.float64 if T == torch.float64 else torch.float32
    s, sp = torch.empty(0, dtype=T).element_size(), torch.empty(0, dtype=P).element_size()
    probe = ea.EmbraceNetMultimodal(DictTrial(hp), "A549", "active_E_vs_inactive_E", "cpu", wl["F"])
    d0, d1, c = probe.FFNN_pre_output_size, probe.CNN_pre_output_size, hp["EMBRACENET_embracement_size"]
    g = torch.Generator(device=device).manual_seed(1)
    r = lambda *sh: torch.rand(*sh, device=device, generator=g)
    x0, x1 = r(B, d0).to(T), r(B, d1).to(T)
    w0, w1 = ((r(c, d0) - 0.5) / d0 ** 0.5).to(P), ((r(c, d1) - 0.5) / d1 ** 0.5).to(P)
    b0, b1 = torch.zeros(c, device=device, dtype=P), torch.zeros(c, device=device, dtype=P)
    cdf0, _ = F.select_prep(torch.tensor([[0.578, 0.422]], device=device), None, B)
    rng = F.RngState(seed=3, step_val=1)
    w0c, w1c = w0.to(T), w1.to(T)
    E = torch.empty(B, c, dtype=T, device=device)
    code = torch.empty(B, c, dtype=torch.uint8, device=device)
    dE = (r(B, c) - 0.5).to(T)
    dX0, dX1 = torch.empty_like(x0), torch.empty_like(x1)
    dW0, dW1 = torch.empty(c, d0, device=device, dtype=P), torch.empty(c, d1, device=device, dtype=P)
    db0, db1 = torch.empty(c, device=device, dtype=P), torch.empty(c, device=device, dtype=P)
    L, ptr, st, code_of = ea._lib.lib(), ea._lib.ptr, ea._lib.stream, ea._lib.DTYPE_CODE[T]
    wsp = torch.empty(1 << 24, dtype=torch.uint8, device=device)

    def fwd():
        ea._lib.check(L.emb_embrace_fwd(ptr(x0), ptr(x1), ptr(w0c), ptr(b0), ptr(w1c), ptr(b1), ptr(cdf0), None, rng.seed,
                                        rng.step_val, None, 0, ptr(E), ptr(code), B, d0, d1, c, code_of, st()), "fwd")

    # the backward as the step runs it: on the pre-masked gradients the classifier head leaves (emb_head_ce_masked) when the
    # fp32 tile GEMM takes the shape (csrc/gemm_jobs.h), else from dE and the code bytes (emb_embrace_bwd)
    masked = bool(L.emb_embrace_bwd_masked_supported(B, d0, d1, c, code_of))
    dD0, dD1 = torch.empty_like(dE), torch.empty_like(dE)

    def bwd():
        F.reduce_defer(True)              # (per stream: event_time_us captures on the graph-capture stream)
        if masked:
            ea._lib.check(L.emb_embrace_bwd_masked(ptr(dD0), ptr(dD1), ptr(x0), ptr(x1), ptr(w0c), ptr(w1c), ptr(dX0), ptr(dX1),
                                                   ptr(dW0), ptr(db0), ptr(dW1), ptr(db1), ptr(wsp), wsp.numel(), B, d0, d1, c,
                                                   code_of, st()), "bwd_masked")
        else:
            ea._lib.check(L.emb_embrace_bwd(ptr(dE), ptr(code), ptr(x0), ptr(x1), ptr(w0c), ptr(w1c), ptr(dX0), ptr(dX1),
                                            ptr(dW0), ptr(db0), ptr(dW1), ptr(db1), ptr(wsp), wsp.numel(), B, d0, d1, c, code_of,
                                            st()), "bwd")
    fwd()
    if masked:
        ea._lib.check(L.emb_embrace_premask(ptr(dE), ptr(code), ptr(dD0), ptr(dD1), B, c, code_of, st()), "premask")
    t_f = event_time_us(fwd, 200)
    # the backward kernel alone (its slab reduction is queued, as in the step, where ONE reduction launch serves the whole
    # backward pass; rocprof's per-kernel average in profiles/ is the cross-check for both numbers)
    try:
        t_b = event_time_us(bwd, 200)
    finally:
        F.reset(all_streams=True)         # the queued reductions are not wanted: drop them, deferral off again
    K = d0 + d1
    bytes_f = s * B * K + s * c * K + 2 * sp * c + 4 * B + s * B * c + B * c
    bytes_b = s * B * c + B * c + s * B * K + s * c * K + s * B * K + sp * c * K + 2 * sp * c
    flops_f, flops_b = 2.0 * B * c * K, 4.0 * B * c * K
    out = {}
    for name, t, by, fl in (("embrace_fwd_kernel", t_f, bytes_f, flops_f), ("embrace_bwd_kernel", t_b, bytes_b, flops_b)):
        ai = fl / by
        ridge = MFMA_PEAK_TFLOPS[wl["dtype"]] * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai < ridge:
            out[name] = dict(bound="hbm", achieved=by / t / 1e3, peak=HBM_PEAK_GBS, unit="GB/s")
        else:
            out[name] = dict(bound="mfma", achieved=fl / t / 1e6, peak=MFMA_PEAK_TFLOPS[wl["dtype"]], unit="TFLOP/s")
        out[name].update(frac=out[name]["achieved"] / out[name]["peak"], us_per_launch=t, algorithmic_bytes=by,
                         algorithmic_flops=fl, traffic=None)
    return out, (d0, d1, c)


def _host_cpu():
    """(model string, physical cores usable by this process)"""
    model, phys, pid, cores = "unknown", set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return model, max(1, min(len(phys) or cores, cores))


def _cpu_rate(wl, threads, warm, steps, runs, cap_s):
    """median over `runs` of samples/s for `steps` train steps (each run time-capped at cap_s, never fewer than 3 steps)"""
    from oracle import ref_step
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    hp, B, Fin = wl["hp"], wl["B"], wl["F"]
    model = ref_step.OracleEmbraceNetMultimodal(hp, Fin)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() > 1:
                torch.nn.init.kaiming_uniform_(p, a=5 ** 0.5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    x1, x2, y = synth_batch(B, Fin, wl["pos"], "cpu", 5)
    x1, x2, y = x1.double(), x2.double(), y.view(-1, 1)
    model.train()
    t0 = time.perf_counter()
    for k in range(warm):
        ref_step.train_step(model, opt, x1, x2, y)
        if k >= 1 and time.perf_counter() - t0 > cap_s:
            break
    rates, done = [], 0
    for _ in range(runs):
        n, t0 = 0, time.perf_counter()
        while n < steps and (n < 3 or time.perf_counter() - t0 < cap_s):
            ref_step.train_step(model, opt, x1, x2, y)
            n += 1
        rates.append(n * B / (time.perf_counter() - t0))
        done = n
    rates.sort()
    return rates[len(rates) // 2], done


def cpu_baseline(wl, seconds):
    """The stock-PyTorch CPU restatement of the reference step (oracle/ref_step.py; fp64 like the reference, incl. its
    per-step loss.item() and sklearn average precision), timed on this box's host cores on a bounded sample:
    (a) the bench workload itself (same model and batch as the GPU line): a 3-step probe per thread setting {8, 32, all
    physical cores}, then 3 runs of 30 steps at the best one (`value` = their median; the other settings keep their probe
    figure -- oversubscribed settings are several times slower and only document that);
    (b) the reference's own CPU-runnable case, BASELINE configs[0] (B = 64, c = 512): 20 warm-up, median of 5 runs of up to
    200 steps (time-capped) with 8 threads."""
    model, phys = _host_cpu()
    prev = torch.get_num_threads()
    settings = sorted({phys, min(32, phys), min(8, phys)})
    ref_wl = dict(WORKLOADS["cfg1"])
    try:
        probe = {th: _cpu_rate(wl, th, 1, 3, 1, 1e9) for th in settings}
        best = max(probe, key=lambda th: probe[th][0])
        step_s = wl["B"] / probe[best][0]
        cap = max(40 * step_s, 0.55 * seconds / 3)              # 30 steps per run always fit; the cap only guards a stall
        main = _cpu_rate(wl, best, 1, 30, 3, cap)
        ref_th = min(8, phys)
        ref_case = _cpu_rate(ref_wl, ref_th, 20, 200, 5, max(1.0, 0.2 * seconds / 5))
    finally:
        torch.set_num_threads(prev)
    return dict(value=main[0], unit="samples/s", cores=best, kind="port", cpu_model=model, physical_cores=phys,
                steps_per_run=main[1], runs=3,
                by_threads={str(th): dict(samples_per_s=(main[0] if th == best else probe[th][0]),
                                          steps_per_run=(main[1] if th == best else probe[th][1])) for th in settings},
                reference_case={"workload": ref_wl["name"], "threads": ref_th, "samples_per_s": ref_case[0],
                                "steps_per_run": ref_case[1], "runs": 5},
                sample=f"bench workload B={wl['B']} fp64: 3-step probe per thread setting {settings}, then the median of 3 runs of "
                       f"{main[1]} steps with {best} threads (1 warm-up); the reference's CPU case B=64 c=512 fp64: median of 5 runs "
                       f"of {ref_case[1]} steps with {ref_th} threads (20 warm-up); incl. per-step loss.item() and sklearn AP as "
                       "the reference's loop")


def single_gpu_rate(ea, wl, dtype, device, steps=60, warmup=10):
    """ms/step of the same graph-captured training step at another precision (single GPU; `extra` figures so that a
    same-precision GPU/CPU ratio exists next to the bf16 headline)."""
    from embracenet_amd import optim, training
    F = ea.functional
    torch.manual_seed(1234)
    model = ea.EmbraceNetMultimodal(DictTrial(wl["hp"]), cell_line="A549", task="active_E_vs_inactive_E", device=device,
                                    in_features_FFNN=wl["F"])
    model = training.prepare_model(model, device, dtype).set_rng("philox", seed=2024, row0=0)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    B = wl["B"]
    x1, x2, y = synth_batch(B, wl["F"], wl["pos"], device, 100)
    in_dt = model.compute_dtype or next(model.parameters()).dtype
    x1, x2 = x1.to(in_dt), x2.to(in_dt)
    counts = torch.zeros(2, dtype=torch.int64, device=device)
    table = ea.metrics.StepTable(1, device)
    loss_slot, conf_slot = table.slot()
    model.train()
    ticks = training.fused_ticks(model, opt, device)
    fused_loss = model.fused_loss_ready(B)

    def step():
        opt.zero_grad(set_to_none=True)
        if fused_loss:
            model.arm_fused_loss(F.FusedLoss(y, counts, False, loss_slot, conf_slot, ticks))
        out = model([x1, x2], is_training=True)
        if fused_loss:
            dlogits = out.detach()
        else:
            _, dlogits = F.weighted_ce_with_grad(out, y, class_counts=counts, confusion=conf_slot, loss_out=loss_slot, ticks=ticks)
        F.reduce_defer(True)
        out.backward(dlogits)
        F.reduce_defer(False)
        opt.step()                                        # sums the queued gradient slabs inside its own launch
        F.reduce_flush()                                  # (nothing left: no launch)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if not bool(torch.isfinite(loss_slot).all()):
        raise SystemExit(f"loss is not finite at {dtype}")
    return dict(ms_per_step=ms, samples_per_s=B / ms * 1e3)


def harness_rate(ea, wl, device, epochs=8):
    """The same workload through the HARNESS instead of one resident batch: training.StepRunner (the loop fit_multimodal runs,
    graph-replayed steps) over data.device_loaders -- a split of 17 distinct batches staged in HBM, the reference's balanced
    batch sampler producing the index lists, ONE row-gather launch per batch (features + sequence + labels), one
    device->host copy of the loss / count table per epoch.  The staging-inclusive, driver-timed rate."""
    from embracenet_amd import data, optim, training
    B, Fin, nb = wl["B"], wl["F"], 17
    npos = max(1, int(round(B * wl["pos"])))
    rows = nb * B
    torch.manual_seed(1234)
    model = ea.EmbraceNetMultimodal(DictTrial(wl["hp"]), cell_line="A549", task="active_E_vs_inactive_E", device=device,
                                    in_features_FFNN=Fin)
    model = training.prepare_model(model, device, wl["dtype"]).set_rng("philox", seed=2024)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    x1, x2, _ = synth_batch(rows, Fin, wl["pos"], device, 7)
    y = torch.zeros(rows, dtype=torch.int64, device=device)
    y[torch.randperm(rows, device=device)[:nb * npos]] = 1     # nb * npos positives: every balanced batch has exactly B rows
    in_dt = model.compute_dtype or next(model.parameters()).dtype
    # ceil(rows / batch_size) + 1 = nb batches (the reference's sampler yields one more than its length, dataprepare.py:442-454)
    loaders = data.device_loaders(x1, x2, y, (rows + nb - 2) // (nb - 1), device, balanced=True, feature_dtype=in_dt)
    runner = training.StepRunner(model, opt, device, graph=True)
    table = ea.metrics.StepTable(nb + 2, device)
    model.train()
    rates, sizes = [], set()
    for ep in range(epochs):
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        for a, b, t in training._pairs(loaders):
            runner.train_step(a, b, t, table)
            n += len(t)
            sizes.add(len(t))
        losses, _ = table.fetch()                             # the one device->host copy of the epoch
        torch.cuda.synchronize()
        rates.append(n / (time.perf_counter() - t0))
    if not all(l == l for l in losses.tolist()):
        raise SystemExit("harness leg: loss is NaN")
    timed = sorted(rates[2:])                                 # the first epochs run eagerly / capture the step graphs
    return dict(samples_per_s=timed[len(timed) // 2], samples_per_s_min=timed[0], samples_per_s_max=timed[-1],
                epochs_timed=len(timed), batches_per_epoch=nb, batch_rows=sorted(sizes), split_rows=rows,
                graphs=len(runner._graphs), what="training.StepRunner(graph=True) over data.device_loaders: balanced sampler, one "
                "gather launch per batch from the HBM-resident split (sequence as byte codes), per-epoch table fetch")


def step_accounting(wl, dims, ms_per_step, stats_csv, traffic_csv=None):
    """Algorithmic FLOPs / HBM bytes (SURVEY 8d formulas, per step at this batch) of every kernel class of the training step,
    next to the time the class takes per step in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/, same binary: sum over the class's kernels of Calls x AverageNs, divided by the profiled steps), and the
    whole-step line from the live ms_per_step.  A class may be several launches (all stored-activation conv blocks of a deep
    stack, the three small linear layers ...); `launches` says how many per step.  Elementwise / pooling kernels are priced by
    bytes, contractions by both; `frac` is against the roof that binds for the precision (bf16: 2.5 PFLOP/s, fp32: 157.3
    TFLOP/s, fp64: 78.6; HBM 8 TB/s)."""
    import csv
    hp, B = wl["hp"], wl["B"]
    s = {"bfloat16": 2, "float32": 4, "float64": 8}[wl["dtype"]]
    bf = wl["dtype"] == "bfloat16"
    peak_tf, peak_gbs = MFMA_PEAK_TFLOPS[wl["dtype"]], HBM_PEAK_GBS
    d0, d1, c = dims
    # conv geometry (CNN_pre.py:24-60): L 256 -> 124 -> 58 -> 25 -> 8
    convs, cin, L = [], 4, 256
    for i in range(hp["CNN_n_layers"]):
        co, k = hp[f"CNN_out_channels_l{i}"], hp[f"CNN_kernel_size_l{i}"]
        Lp = (L - 10) // 2 + 1
        convs.append(dict(cin=cin, cout=co, k=k, L=L, Lp=Lp, flops=2.0 * cin * co * k * L * B,
                          x=B * L * max(cin, 8) * s, y=B * L * co * s, pooled=B * Lp * co * s, arg=B * Lp * co))
        cin, L = co, Lp
    ffw = [wl["F"]] + [hp[f"FFNN_n_units_l{i}"] for i in range(hp["FFNN_n_layers"])]
    ffnn = sum(2.0 * a * b_ for a, b_ in zip(ffw[:-1], ffw[1:])) * B
    pw = [c] + [hp[f"EMBRACENET_n_units_l{i}"] for i in range(hp["n_post_layers"])]
    post = sum(2.0 * a * b_ for a, b_ in zip(pw[:-1], pw[1:])) * B          # hidden post layers (the head is its own class)
    K = d0 + d1
    c0, rest = convs[0], convs[1:]
    emb_f = (2.0 * B * c * K, s * B * K + s * c * K + s * B * c + B * c)
    emb_b = (4.0 * B * c * K, s * B * c + B * c + 2 * s * B * K + s * c * K + 4 * c * K)
    rider = bf      # bf16: the epigenomic MLP rides on first_stats / bn_bwd_dz (csrc/rider.h), whose lines carry its work
    # class -> (kernel-name fragments, flops per step, bytes per step)
    table = {}
    if bf:
        table.update({
            "first_stats": (["first_stats_rider_kernel", "first_kernel<2, 2, 0>"], c0["flops"] + ffnn, B * 4 * 256 * s + c0["x"] + B * wl["F"] * s),
            "first_apply": (["first_kernel<2, 2, 1>"], c0["flops"], c0["x"] + c0["pooled"] + c0["arg"]),
            "first_bwd_acc": (["first_kernel<2, 2, 4>"], c0["flops"], c0["x"] + c0["pooled"] + c0["arg"]),
            "first_bwd_sums": (["first_kernel<2, 2, 2>"], c0["flops"], c0["x"] + c0["pooled"] + c0["arg"]),
            "first_bwd_wgrad": (["first_kernel<2, 2, 3>"], 2 * c0["flops"], c0["x"] + c0["pooled"] + c0["arg"]),
        })
        if len(convs) > 1:
            c1 = convs[1]
            table.update({
                "conv2_fwd": (["conv_t_stream_kernel<2, 2, true", "conv_t_stream_kernel<4, 2, true", "conv_t_kernelIDF16bLi2"], c1["flops"], c1["x"] + c1["y"]),
                "conv2_dgrad": (["conv_t_stream_kernel<4, 1, false", "conv_t_kernelIDF16bLi4"], c1["flops"], c1["y"] + c1["x"]),
                "conv2_wgrad": (["conv_wgrad_stream", "conv_wgrad_direct"], c1["flops"], c1["y"] + c1["x"]),
                "conv2_wgrad+dgrad": (["conv_bwd_dual_kernel"], 2 * c1["flops"], 2 * (c1["y"] + c1["x"])),
            })
    else:
        # fp32 / fp64: block 1 and the stored-activation blocks run the same kernel families; all convolution launches of the
        # step are ONE class (forward + input gradient of blocks >= 2 + weight gradient of every block), the tile GEMM launches
        # of csrc/gemm_jobs.h carry the fusion layer's backward and the backward of large Linear layers as well
        conv_fl = 3.0 * sum(cv["flops"] for cv in convs) - c0["flops"]
        conv_by = sum(3 * (cv["x"] + cv["y"]) for cv in convs)
        table["convolutions (all blocks: fwd, dgrad, wgrad)"] = (["conv_t_kernel<", "conv_direct_kernel<", "conv_wgrad_direct_kernel<", "conv_gemm_kernel",
                                                                  "conv_wgrad_kernel<", "first_kernel<"], conv_fl, conv_by)
        table["fp32 tile GEMM (conv blocks with >= 128 channels, fusion backward, large Linear backward)"] = (["gemm_jobs_kernel"], 0.0, 0)   # priced below
    table.update({
        "embrace_fwd": (["embrace_fwd"], *emb_f),
        "embrace_bwd": (["embrace_bwd"], *emb_b),
        "head_ce": (["head_ce"], 3 * 2.0 * B * pw[-1] * 2, 2 * s * B * pw[-1]),
        "bn_relu_pool": (["bn_relu_pool"], 0.0, sum(cv["y"] + cv["pooled"] + cv["arg"] for cv in (rest if bf else convs))),
        "bn_bwd_dz": (["bn_bwd_dz"], (2 * ffnn if rider else 0.0), sum(cv["pooled"] + cv["arg"] + 2 * cv["y"] for cv in (rest if bf else convs)) + (B * wl["F"] * s if rider else 0)),
        "bn_bwd_affine": (["bn_bwd_affine"], 0.0, sum(3 * cv["y"] for cv in (rest if bf else convs))),
        "optimizer (+ slab sums)": (["multi_opt_kernel"], 0.0, 0),
    })
    if not rider:
        table["mlp / linear layers (epigenomic stack, hidden post layers)"] = (["mlp_fwd", "mlp_bwd", "linear_fwd_kernel", "linear_bwd_kernel"],
                                                                            3 * (ffnn + post), 3 * B * (wl["F"] + c) * s)
    elif post:
        table["hidden post layers"] = (["mlp_fwd", "mlp_bwd", "linear_fwd_kernel", "linear_bwd_kernel"], 3 * post, 3 * B * c * s)
    rows, steps = {}, 1.0
    if stats_csv and os.path.exists(stats_csv):
        for r in csv.DictReader(open(stats_csv)):
            rows[r["Name"]] = (float(r["Calls"]), float(r["AverageNs"]) / 1e3)
        once = [v[0] for n, v in rows.items() if "multi_opt_kernel" in n or "head_ce_kernel" in n or "weighted_ce" in n]
        steps = max(1.0, min(once)) if once else 1.0
    if rows and not bf:
        # the tile GEMM launches: the fusion backward (its class is then empty), the conv jobs they took over and the backward of
        # Linear layers with B * in * out >= 2^27 (csrc/linear.hip)
        ring = [n for n in rows if "gemm_jobs_kernel" in n]
        if ring:
            ring_conv = sum(cv["flops"] * (2 if (cv["cin"] % 32 == 0 and cv["cout"] >= 128) else 0) for cv in rest)      # forward + weight gradient
            ring_conv += sum(cv["flops"] for cv in rest if cv["cout"] % 32 == 0 and cv["cin"] >= 128)                    # input gradient
            big_lin = [(a, b_) for a, b_ in list(zip(ffw[:-1], ffw[1:])) + list(zip(pw[:-1], pw[1:]))
                       if B * a * b_ >= (1 << 27) and a % 4 == 0 and b_ % 4 == 0]
            lin_bwd = sum(4.0 * B * a * b_ for a, b_ in big_lin)
            name = "fp32 tile GEMM (conv blocks with >= 128 channels, fusion backward, large Linear backward)"
            table[name] = (["gemm_jobs_kernel"], ring_conv + emb_b[0] + lin_bwd,
                           emb_b[1] + sum(2 * (cv["x"] + cv["y"]) for cv in rest if cv["cout"] >= 128) + sum(s * (2 * B * (a + b_) + 2 * a * b_) for a, b_ in big_lin))
            cname = "convolutions (all blocks: fwd, dgrad, wgrad)"
            f_, fl, by = table[cname]
            table[cname] = (f_, fl - ring_conv, by)
            lname = "mlp / linear layers (epigenomic stack, hidden post layers)"
            if lname in table and lin_bwd:
                f_, fl, by = table[lname]
                table[lname] = (f_ + ["linear_premask_kernel"], fl - lin_bwd, by)
            table.pop("embrace_bwd", None) if not any("embrace_bwd" in n for n in rows) else None
        else:
            table.pop("fp32 tile GEMM (conv blocks with >= 128 channels, fusion backward, large Linear backward)", None)
    if rows:   # classes whose kernels did not run in this step
        table = {cls: v for cls, v in table.items() if any(f in n for f in v[0] for n in rows)}
    hbm = {}    # kernel name -> HBM bytes per launch from the committed PMC passes of this command (tools/run_pmc_hbm.sh)
    if traffic_csv and os.path.exists(traffic_csv):
        for r in csv.DictReader(open(traffic_csv)):
            hbm[r["kernel"]] = float(r["hbm_MB_per_launch"]) * 1e6
    kernels = {}
    for cls, (frags, fl, by) in table.items():
        hit = [(n, v) for n, v in rows.items() if any(f in n for f in frags)]
        us = sum(calls * avg for _, (calls, avg) in hit) / steps if hit else None
        ent = dict(algorithmic_flops=fl, algorithmic_bytes=by, us_per_step=us, launches=round(sum(c_ for _, (c_, _a) in hit) / steps, 2) if hit else None)
        tr = [v for f in frags for n, v in hbm.items() if f in n]
        if tr and by:
            ent.update(traffic=sum(tr), traffic_over_algorithmic=sum(tr) / by)
        if us and (fl or by):
            tf, gbs = fl / us / 1e6, by / us / 1e3
            ent.update(tflops=tf, gbs=gbs, frac=max(tf / peak_tf, gbs / peak_gbs), bound="mfma" if tf / peak_tf >= gbs / peak_gbs else "hbm")
        kernels[cls] = ent
    total_fl = 3.0 * (sum(cv["flops"] for cv in convs) + ffnn + post + 2.0 * B * c * K + 2.0 * B * pw[-1] * 2)
    total_by = sum(by for _, _, by in table.values())
    us_step = ms_per_step * 1e3
    return dict(kernels=kernels, source=os.path.basename(stats_csv) if rows else None, profiled_steps=steps if rows else None,
                traffic_source=os.path.basename(traffic_csv) if hbm else None,
                whole_step=dict(algorithmic_flops=total_fl, algorithmic_bytes=total_by, us=us_step, tflops=total_fl / us_step / 1e6,
                                gbs=total_by / us_step / 1e3, frac_mfma=total_fl / us_step / 1e6 / peak_tf,
                                frac_hbm=total_by / us_step / 1e3 / peak_gbs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, help="override the workload's precision")
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture")
    ap.add_argument("--backend", default=None, help="nccl (RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=120.0)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline and cpu_baseline legs")
    ap.add_argument("--roofline-only", action="store_true", help="only time the isolated kernels (used under rocprofv3 --pmc)")
    ap.add_argument("--packed-input", action="store_true",
                    help="stage the DNA windows as uint8 base codes (SURVEY 8 row f4) instead of the loader's [B,4,256] floats")
    ap.add_argument("--no-fused-loss", action="store_true", help="A/B: head, loss and head backward as three launches")
    ap.add_argument("--sync-bn", action="store_true", help="N>1: BatchNorm statistics of the global batch (parity switch, "
                    "dist.set_sync_batchnorm); default is local statistics")
    ap.add_argument("--split-graph", action="store_true", help="N>1: keep the all-reduce outside the captured graphs")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the N>1 code path (flat gradient buffer, all-reduce in the step) with a single rank")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; ms_per_step is their median")
    ap.add_argument("--no-harness", action="store_true", help="skip the extra.harness leg (StepRunner over device_loaders)")
    args = ap.parse_args()

    import embracenet_amd as ea
    from embracenet_amd import dist as D, optim, training
    rank, local_rank, world = D.init(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    one_gpu = args.backend == "gloo"
    device = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(device)
    wl = dict(WORKLOADS[args.workload])
    if args.dtype:
        wl["dtype"] = args.dtype
    B, Fin = wl["B"], wl["F"]
    dist_path = world > 1 or args.force_collectives      # flat gradient buffer + all-reduce inside the step
    if args.force_collectives and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group(args.backend or "nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1)
    D.FORCE_COLLECTIVES = bool(args.force_collectives)

    if args.roofline_only:
        kern, dims = kernel_roofline(ea, wl, device)
        print(json.dumps({"kernels": kern, "shapes": dims}), flush=True)
        return

    torch.manual_seed(1234)                               # identical initial weights on every rank
    model = ea.EmbraceNetMultimodal(DictTrial(wl["hp"]), cell_line="A549", task="active_E_vs_inactive_E", device=device,
                                    in_features_FFNN=Fin)
    model = training.prepare_model(model, device, wl["dtype"]).set_rng("philox", seed=2024, row0=rank * B)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    if args.sync_bn:
        D.set_sync_batchnorm(model)
    x1, x2, y = synth_batch(B, Fin, wl["pos"], device, 100 + rank)
    in_dt = model.compute_dtype or next(model.parameters()).dtype   # inputs staged in the compute dtype
    x1, x2 = x1.to(in_dt), (ea.functional.pack_onehot(x2) if args.packed_input else x2.to(in_dt))
    counts = torch.zeros(2, dtype=torch.int64, device=device)
    table = ea.metrics.StepTable(1, device)
    loss_slot, conf_slot = table.slot()
    F = ea.functional
    model.train()
    ticks = training.fused_ticks(model, opt, device)      # RNG step + optimizer step advance inside the loss kernel
    # N > 1: every gradient is a view of one of TWO flat buffers that the backward kernels write directly (the same
    # dist.BucketedFlatGrads the harness's StepRunner uses): the bucket of the fusion layer and the head is all-reduced
    # asynchronously as soon as the fusion backward is enqueued -- behind ~100 us of pre-network backward --, the
    # pre-network bucket after the backward.  Two trailing slots of the second bucket carry the NEXT step's local
    # (positives, rows): the global class counts a step needs before its loss are thus reduced one step ahead
    # inside the gradient collective (labels are known when a batch is staged).
    fused_loss = (not args.no_fused_loss) and model.fused_loss_ready(B)
    # fused head: the count exchange needs no launches of its own -- the head reads the global counts from, and writes the
    # shard's counts into, a four-float block that lives in the late gradient bucket (csrc/head.hip, global_counts = 2);
    # one two-element copy after the all-reduce moves the sums into place for the next step
    exch = dist_path and fused_loss
    flat = D.BucketedFlatGrads(model, extra=4 if exch else 2) if dist_path else None
    local_counts = F.count_labels(y).to(torch.float32) if (dist_path and not exch) else None

    overlap = {"early": True}   # first bucket's all-reduce issued from inside the backward (off when the backward is captured
                                # in a graph that must not contain collectives)

    def fwd_bwd():
        if flat is None:
            opt.zero_grad(set_to_none=True)
        if fused_loss:                                    # the head's launch takes the loss and its own backward along
            model.arm_fused_loss(F.FusedLoss(y, flat.extra if exch else counts, 2 if exch else dist_path, loss_slot, conf_slot, ticks))
        out = model([x1, x2], is_training=True)
        if fused_loss:
            dlogits = out.detach()                        # ignored by the head's node (the armed loss is the graph's root)
        else:
            _, dlogits = F.weighted_ce_with_grad(out, y, class_counts=counts, global_counts=dist_path, confusion=conf_slot,
                                                 loss_out=loss_slot, ticks=ticks)
        F.reduce_defer(True)                              # the weight-gradient slabs of the backward are queued: one process ->
        if dist_path and overlap["early"]:                # the optimizer launch sums them; N > 1 -> reduced before each
            def early():                                  # all-reduce, which needs finished gradients
                F.reduce_flush()                          # (head + fusion slabs)
                flat.allreduce_early()
            F.set_after_embrace_backward(early)
        try:
            out.backward(dlogits)
        finally:
            F.set_after_embrace_backward(None)
            F.reduce_defer(False)
        if dist_path:
            F.reduce_flush()                              # (pre-network slabs)

    def reduce_grads():
        if exch:
            flat.finish()                                 # waits for the first bucket, reduces the second (+ the shards' counts)
            F.park_copy(flat.extra[2:4], flat.extra[0:2])   # done by the optimizer launch that follows
            return
        flat.extra.copy_(local_counts)                    # next batch's labels (synthetic: the same batch)
        flat.finish()
        counts.copy_(flat.extra.round().to(torch.int64))

    if dist_path:                                         # counts of the very first step
        F.count_labels(y, out=counts)
        D.allreduce_counts(counts)
        if exch:
            flat.extra.zero_()
            flat.extra[0:2] = counts.to(torch.float32)

    def opt_step():
        opt.step()
        if not dist_path:
            F.reduce_flush()                              # whatever the optimizer launch did not take (nothing: no launch)

    def eager_step():
        fwd_bwd()
        if flat is not None:
            reduce_grads()
        opt_step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                         # builds optimizer state, MIOpen plans, LDS attributes
        for _ in range(3):
            eager_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    use_graph = not args.eager
    if use_graph and args.sync_bn and dist_path and torch.distributed.get_backend() != "nccl":
        use_graph = False                                 # the BatchNorm exchanges sit inside forward/backward: only RCCL
        print("[bench] --sync-bn on a non-RCCL backend: collectives cannot be captured, running eager steps",   # captures
              file=sys.stderr, flush=True)
    if use_graph:
        graph_mode = "one hipGraph per step"
        step = None
        if not dist_path:
            g_all = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_all):
                fwd_bwd()
                opt_step()
            step = g_all.replay
        elif not args.split_graph and torch.distributed.get_backend() == "nccl":
            # RCCL collectives are capturable: the whole step, all-reduce included, replays as ONE graph (no host
            # round trip between backward, collective and optimizer).  Any failure falls back to the split form.
            try:
                g_all = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_all):
                    fwd_bwd()
                    reduce_grads()
                    opt_step()
                g_all.replay()
                torch.cuda.synchronize()
                step = g_all.replay
                graph_mode = "one hipGraph per step, RCCL all-reduce captured"
            except Exception as e:                        # noqa: BLE001 - report and fall back
                if rank == 0:
                    print(f"[bench] single-graph capture with the collective failed ({type(e).__name__}: {e}); "
                          "using two graphs around an eager all-reduce", file=sys.stderr, flush=True)
                step = None
        if step is None:                                  # collectives stay outside the captured regions
            overlap["early"] = False
            g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_fb):
                fwd_bwd()
            with torch.cuda.graph(g_opt, pool=g_fb.pool()):
                opt_step()
            graph_mode = "two hipGraphs around an eager all-reduce"

            def step():
                g_fb.replay()
                reduce_grads()
                F.flush_copy()                            # (the captured optimizer launch cannot take a copy parked after its capture)
                g_opt.replay()
    else:
        step = eager_step
        graph_mode = "eager"

    for _ in range(args.warmup):
        step()
    # `--windows` timed windows of EXACTLY --steps steps each, every one bracketed by synchronize + barrier on both sides and
    # taken as the MAX over ranks; ms_per_step / value come from the MEDIAN window (one 20-step window of this step is 4 ms
    # -- too thin against host jitter), every window is listed in extra.windows_ms_per_step
    windows = []
    for _ in range(max(1, args.windows)):
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        windows.append(D.max_over_ranks(time.perf_counter() - t0, device))
    elapsed = sorted(windows)[len(windows) // 2]
    final_loss = float(loss_slot.item())
    devices = [None] * world                                # which card every rank really ran on
    me = dict(rank=rank, device=str(device), name=torch.cuda.get_device_name(device), pid=os.getpid())
    if world > 1:
        torch.distributed.all_gather_object(devices, me)
    else:
        devices = [me]
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")

    result = None
    if rank == 0:
        value = world * B * args.steps / elapsed
        result = {
            "metric": "training samples/sec (EmbraceNet, A549 enhancers), whole job; per GPU = value / n_gpus",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bfloat16": "bf16", "float32": "f32", "float64": "f64"}[wl["dtype"]], "data": "synthetic",
            "per_gpu": value / world,
            "config": {"workload": wl["name"], "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world} (batch-sharded, all-reduce of gradients)" if world > 1 else "single GPU",
                       "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                       "ranks_seen": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
                       "rank_devices": devices, "self_launched": bool(os.environ.get("EMB_BENCH_LAUNCHED")),
                       "timing": f"median of {len(windows)} windows of {args.steps} steps (max over ranks per window)",
                       "loss": "inside the classifier-head launch" if fused_loss else "own launch", "step": "zero_grad+fwd+weighted CE+bwd" + ("+allreduce" if world > 1 else "") + "+fused Adam",
                       "graph": bool(use_graph), "graph_mode": graph_mode,
                       "sequence_input": "uint8 base codes [B,256]" if args.packed_input else "one-hot [B,4,256] (loader format)", "rng": "philox (device-side modality dropout and selection)",
                       "batchnorm": "global-batch statistics (all-reduced sums)" if (args.sync_bn and D.collectives_on()) else "local statistics per rank",
                       "final_loss": final_loss},
        }
        result["extra"] = {"windows_ms_per_step": [1e3 * w / args.steps for w in windows],
                           "ms_per_step_min": 1e3 * min(windows) / args.steps, "ms_per_step_max": 1e3 * max(windows) / args.steps}
        if world == 1 and not args.no_extras:
            kern, dims = kernel_roofline(ea, wl, device)
            dom = max(kern, key=lambda k: kern[k]["us_per_launch"])
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                for k, v in json.load(open(pmc)).get(args.workload, {}).items():
                    if k in kern:
                        kern[k]["traffic"] = v
            roof = dict(kern[dom])
            roof.update(kernel=dom, shapes=dict(B=B, d0=dims[0], d1=dims[1], c=dims[2]), kernels=kern)
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"bench_{args.workload}_kernel_stats.csv"))
            pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_hbm_per_kernel.csv"))
            roof["step"] = step_accounting(wl, dims, result["ms_per_step"], os.path.join(ROOT, "profiles", prof[-1]) if prof else None,
                                           os.path.join(ROOT, "profiles", pmcs[-1]) if pmcs and args.workload == "cfg2" else None)
            result["roofline"] = roof
            extra = {}
            for dt in ("float32", "float64"):                     # the same step at the reference's precision and at fp32
                if dt != wl["dtype"]:
                    extra[dt] = single_gpu_rate(ea, wl, dt, device)
            result["extra"]["gpu_step_other_precisions"] = extra
            if not args.no_harness:
                result["extra"]["harness"] = harness_rate(ea, wl, device)
            result["cpu_baseline"] = cpu_baseline(wl, args.cpu_baseline_seconds)
            result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
            if "float64" in extra:
                result["extra"]["fp64_gpu_vs_fp64_cpu"] = extra["float64"]["samples_per_s"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
