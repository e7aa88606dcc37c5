#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec of one full discriminator + generator training step
(/root/reference/model_wrapper.py:131-190 semantics incl. both Adam steps) on synthetic 256x256 batches.

    python bench.py --gpus N --steps K --warmup W

N > 1: the script starts its own N ranks (one process per GPU, spawned BEFORE the parent touches the GPU; the parent only
relays rank 0's JSON line) unless it already runs under ``python -m torch.distributed.run`` (RANK / WORLD_SIZE set), in which
case it is a rank.  Rendezvous is 127.0.0.1.

Workload (BASELINE.json metric): channel_factor=1, batch 20 per GPU, bf16 storage / bf16 MFMA / fp32 accumulate,
synthetic images / labels / masks with the reference's input contract, random-init G and D, kaiming-init frozen
VGG-16, Adam lr 1e-5.  The LAST stdout line of rank 0 is ONE compact JSON record (compact_line(): < 4 KB, tests/test_bench_line.py);
the full record (per-route tables, families, notes, parity sub-objects) goes to bench_detail.json beside this script and to stderr.
`value` is the K timed steps of the contract, taken right after the >= 6 s `sustained` window on the same job (clocks settled).
Beside the headline the full record carries (N = 1 only):
  roofline            achieved / frac: the dominant kernel; conv_frac: FLOP-weighted over every convolution launch (families: forward /
                      input gradient / weight gradient, event-timed on the launch stream in eager steps); sn3x3_bwd: the north-star's
                      3x3 spectral-norm backward; step_frac: the whole step on EXECUTED FLOPs; nonconv_floor_ms measured in the run
  channel_factor2/0.5 BASELINE.json config 4 on one GPU, each with its own probe;  fp16: the half-precision storage mode + its parity
  parity_mode         the same step in the fp32 mode (exact-fp32 MFMA, ordered reductions) - the mode that carries the
                      1e-3 parity contract (tests/test_gpu_step.py)
  batch32             BASELINE.json config 2 (one GPU, bf16, batch 32)
  cpu_baseline        the CPU oracle (torch fp32 on the host cores) at batch 2 and batch 20, with the CPU model
"""
import argparse
import gc
import json
import re
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FLOPs per image of the full step (2 x MACs of every conv / linear / bmm), necessary work only
# (SURVEY.md section 8d): D-step 210.40 + G-step 199.34 GFLOP at channel_factor = 1.
GFLOP_PER_IMAGE = {1: 409.74, 2: 197.75, 0.5: 1239.60}
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}
DOMINANT_KERNEL = "conv3x3_pp_kernel<bf16,2> (sp_conv2d_igemm, 3x3, Cout>64, 128 co x 8x32 px tiles, ping-pong schedule)"
DOMINANT_KERNEL_SYMBOL = r"conv3x3_pp_kernel<bf16, 2, [^>]*, 2(, false)?(, (false|true))?>\("      # regex: every epilogue form on 32-wide tiles (not the 16-wide form; optional trailing arguments: round 4's fused-tail flag - off - and round 5's window-position flag)
TRAFFIC_FILES = ("round6_hbm_traffic_per_kernel.json", "round5_hbm_traffic_per_kernel.json", "round4_hbm_traffic_per_kernel.json", "round3_hbm_traffic_per_kernel.json", "round2_hbm_traffic_per_kernel.json")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=20, help="batch per GPU")
    p.add_argument("--channel-factor", type=float, default=1)
    p.add_argument("--dtype", choices=("bf16", "f32", "fp16"), default="bf16",
                   help="storage type of activations / packed weights: bf16 (headline, BASELINE config 2), f32 (parity mode), fp16 "
                        "(BASELINE config 5's activations; static loss scale)")
    p.add_argument("--fp8", action="store_true", help="BASELINE config 5: e4m3 operands on the fp8 MFMA for the VGG-16 pyramid's wide 3x3 "
                                                      "layers in the no-gradient pass (ops.set_vgg_fp8(1)); with --dtype fp16 this is the config-5 line")
    p.add_argument("--deterministic", action="store_true",
                   help="SP_TUNE_DETERMINISTIC=1: every reduction in a fixed order also in the 16-bit modes (no fp32 atomics in the small-map "
                        "weight gradients): two runs give identical bits; the default line reports this mode's cost as `deterministic`")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-probe", action="store_true")
    p.add_argument("--no-sub-records", action="store_true", help="skip the fp32 parity-mode and batch-32 sub-records")
    p.add_argument("--device-masks", action="store_true",
                   help="draw a fresh batch of training masks ON THE DEVICE every step (synthetic.training_masks_device, SURVEY.md "
                        "row f1) inside the timed region instead of reusing one resident batch")
    p.add_argument("--no-graphs", action="store_true",
                   help="launch every kernel eagerly; default: the D phase and the G phase are replayed as captured HIP "
                        "graphs (ModelWrapper.capture_graphs), which keeps a slow host from becoming the bottleneck")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without an outer launcher: spawn the ranks (the parent never initialises the GPU)
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int, timeout_s: float = None) -> int:
    """Starts the n ranks as fresh children (never re-executes a process that touched the GPU), relays rank 0's stdout, and
    supervises them: when any rank exits non-zero, or nothing finishes within `timeout_s` (default: BENCH_RANK_TIMEOUT_S or
    1800 s - a rank stuck at the rendezvous or in a collective whose partner died), the remaining ranks are terminated (then
    killed) and the worst exit code is returned.  A hung multi-GPU run thus ends with a message instead of blocking the driver."""
    import socket
    import tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get("BENCH_RANK_TIMEOUT_S", "1800"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")             # rank 0's stdout (a pipe would need a reader thread to stay drained)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(globals()["__file__"])] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    deadline = time.time() + timeout_s
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = bad[0][1], "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc, why = 124, "no completion within %.0f s (ranks still running: %s)" % (timeout_s, [r for r, c in enumerate(codes) if c is None])
            break
        time.sleep(0.2)
    if why is not None:
        print("bench.py: %s; stopping the other ranks" % why, file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()                            # exact PIDs this function started
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (reported beside the GPU number, never the thing measured as `value`)
# ------------------------------------------------------------------------------------------------------------------
def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cf, seconds_budget=30.0):
    """The CPU oracle (torch fp32) on a bounded sample of the same workload: batch 2 (the reference's own CPU-runnable case,
    BASELINE.json configs[0]) and batch 20 (the benchmark's per-GPU batch), SURVEY.md section 8d."""
    from oracle import sempyr_oracle as O
    from semantic_pyramid_for_image_generation_amd import params, synthetic
    # host cores available to this process; beyond ~16 threads oneDNN's conv backward stops scaling on the pool's hosts
    # (a 256-thread run took 410 s per batch-2 step), so the thread count - reported as `cores` - is capped there
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    t_start = time.time()
    results = {}
    ref20 = None
    for b, max_steps, budget in ((2, 4, 0.3 * seconds_budget), (20, 2, seconds_budget)):
        # fresh networks per batch size (synthesised from (seed, key, shape): cheap), so that the FIRST batch-20 step is a known
        # function of (parameters seed 0/1/2, batch seed 0, latent seed 20) - the GPU repeats exactly that step (parity_b20)
        G = O.make_state(params.synth_state_dict(O.layout_template(O.generator_layout(cf)), 0))
        D = O.make_state(params.synth_state_dict(O.layout_template(O.discriminator_layout(cf)), 1))
        V = O.make_state(params.synth_state_dict(O.layout_template(O.vgg16_layout()), 2), frozen=True)
        og = torch.optim.Adam(O.trainable(G), lr=1e-5)
        od = torch.optim.Adam(O.trainable(D), lr=1e-5)
        g = torch.Generator().manual_seed(b)
        images, labels, masks = synthetic.synthetic_batch(b, 0)
        times = []
        for i in range(max_steps):
            nd, ng = torch.randn(b, 128, generator=g), torch.randn(b, 128, generator=g)
            t0 = time.time()
            out = O.train_step(G, D, V, og, od, images, labels, masks, nd, ng, skip_dead_d_wgrad=True)
            times.append(time.time() - t0)
            if b == 20 and i == 0:
                ref20 = {k: float(out[k]) for k in ("loss_d_real", "loss_d_fake", "loss_g", "loss_rec", "loss_div")}
                ref20["pixels"] = out["images_fake_g"].detach().float().contiguous().flatten()[parity_b20_index()].clone()
            if time.time() - t_start > budget and len(times) >= 2:
                break
        steady = min(times[1:]) if len(times) > 1 else times[0]
        results[b] = (b / steady, len(times), steady)
        del G, D, V, og, od
    avail_note = ("threads capped at 16 of %d: beyond that oneDNN's convolution backward stops scaling on this host "
                  "(a 256-thread batch-2 step took 410 s); " % avail) if avail > 16 else ""
    return {"value": round(results[2][0], 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_name(), "host_cores_available": avail, "steps_b2": results[2][1],
            "batch20": {"value": round(results[20][0], 4), "unit": "images/sec", "steps": results[20][1], "step_s": round(results[20][2], 2)},
            "sample": avail_note + "oracle/sempyr_oracle.py (torch fp32 CPU restatement, pinned to the reference goldens), cf=%g, 256x256: "
                      "%d steps of batch 2 (best steady step %.2fs) -> value; %d steps of batch 20 (best %.2fs) -> batch20"
                      % (cf, results[2][1], results[2][2], results[20][1], results[20][2]),
            "_ref20": ref20}


def parity_b20_index():
    return torch.randint(0, 20 * 3 * 256 * 256, (4096,), generator=torch.Generator().manual_seed(2020))


def parity_b20_record(dev, ref20, cf=1):
    """Full-size end-to-end comparison (round-3 VERDICT, missing #4): the FIRST batch-20 step of the CPU oracle that cpu_baseline
    just timed (parameters synthesised from seeds 0 / 1 / 2, batch seed 0, latents from seed 20) repeated by the HIP path in its
    fp32 parity mode: the five losses and 4096 generator pixels.  tests/test_gpu_configs.py holds the same comparison to 1e-3."""
    import semantic_pyramid_for_image_generation_amd as sp
    from semantic_pyramid_for_image_generation_amd import ops, params, synthetic
    from oracle import sempyr_oracle as O
    ops.set_compute_dtype(torch.float32)
    try:
        G, D, V = sp.Generator(channels_factor=cf), sp.Discriminator(channel_factor=cf), sp.VGG16()
        G.load_state_dict(params.synth_state_dict(O.layout_template(O.generator_layout(cf)), 0))
        D.load_state_dict(params.synth_state_dict(O.layout_template(O.discriminator_layout(cf)), 1))
        V.load_state_dict(params.synth_state_dict(O.layout_template(O.vgg16_layout()), 2))
        G.to(dev).train(); D.to(dev).train(); V.to(dev).eval()
        mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=torch.optim.Adam(G.parameters(), lr=1e-5),
                             discriminator_optimizer=torch.optim.Adam(D.parameters(), lr=1e-5), save_data_path=None)
        images, labels, masks = synthetic.synthetic_batch(20, 0)
        g = torch.Generator().manual_seed(20)
        nd, ng = torch.randn(20, 128, generator=g), torch.randn(20, 128, generator=g)
        out = mw.train_step(images.to(dev), labels.to(dev), [m.to(dev) for m in masks], noise_d=nd.to(dev), noise_g=ng.to(dev))
        pairs = (("loss_discriminator_real", "loss_d_real"), ("loss_discriminator_fake", "loss_d_fake"), ("loss_generator", "loss_g"),
                 ("loss_generator_semantic_reconstruction", "loss_rec"), ("loss_generator_diversity", "loss_div"))
        loss_err = max(abs(float(out[a]) - ref20[r]) / max(abs(ref20[r]), 2e-2) for a, r in pairs)
        pix = out["images_fake"].float().cpu().contiguous().flatten()[parity_b20_index()]
        err = (pix - ref20["pixels"]).abs()
        return {"against": "the CPU oracle's first batch-20 step of this run (cf=%g, parameters seeds 0/1/2, batch seed 0)" % cf,
                "mode": "fp32 storage, exact-fp32 MFMA", "worst_loss_rel_err": round(loss_err, 7),
                "worst_pixel_abs_err": round(float(err.max()), 7), "pixel_rms_err": round(float((err ** 2).mean().sqrt()), 8), "bound": 1e-3}
    finally:
        ops.set_compute_dtype(torch.bfloat16)
        gc.collect()
        torch.cuda.empty_cache()


def recorded_traffic(symbols):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected in
    separate runs of this script and reduced by profiles/extract_traffic.py); PMC counters cannot be read from inside the
    process, so the committed summary is reported WITH ITS PROVENANCE (file, sha256 of the file, the kernel name matched) -
    (None, None) if it is missing."""
    import hashlib
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            raw = open(path, "rb").read()
            kernels = json.loads(raw)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for symbol in symbols:
            # every instantiation of the kernel that serves the dominant launches (e.g. its two epilogue forms), launch-weighted
            hits = [(kname, rec) for kname, rec in kernels.items() if re.search(symbol, kname)]
            if hits:
                n = sum(rec.get("launches_profiled") or 1 for _, rec in hits)
                avg = sum(rec["hbm_bytes_per_launch"] * (rec.get("launches_profiled") or 1) for _, rec in hits) / n
                return int(avg), {"file": "profiles/" + name, "sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                                  "kernel": " + ".join(k[:96] for k, _ in hits), "launches_profiled": n}
    return None, None


def is_dominant_route(route: str) -> bool:
    """Launches served by the dominant kernel: conv3x3_pp_kernel<16bit, 2, ..., FW = 2> in every epilogue form (plain, pooled, pooled with
    window positions) - the set DOMINANT_KERNEL_SYMBOL matches in a rocprofv3 trace; not its 16-pixel-wide tiles (",w16"), not the
    64-channel form (<16bit,1,...>).  fp32 mode: the tall kernel on the same tiles."""
    return (route.startswith("conv3x3_pp<16bit,2") and "w16" not in route) or route == "conv3x3_tall<f32,2,8>"


_SLEEP_CYCLES_PER_MS = None


def gpu_blocker(ms: float) -> None:
    """Keeps the stream busy for ~ms (torch.cuda._sleep, calibrated once with events): everything enqueued behind it waits, so the
    host gets a head start of that length over the GPU."""
    global _SLEEP_CYCLES_PER_MS
    if _SLEEP_CYCLES_PER_MS is None:
        torch.cuda._sleep(100000)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(4000000)
        e1.record()
        torch.cuda.synchronize()
        _SLEEP_CYCLES_PER_MS = 4000000 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms * _SLEEP_CYCLES_PER_MS))


def kernel_probe(step_fn, peak, steps=6, step_ms=None):
    """Every convolution launch (forward, input gradient, weight gradient - 99.6 % of the step's FLOPs) inside `steps` extra
    EAGER training steps is bracketed by events on the launch stream; algorithmic FLOPs of a launch = 2*M*N*K.

    Round-3 VERDICT (weak #3): with the host behind the GPU an event interval contains launch latency, and the figures of a slow
    host were irreproducible.  Now (a) every probe step is enqueued behind a blocker that holds the stream for longer than the
    host needs to enqueue the whole step, so the queue never runs dry; (b) a launch's time is its MEDIAN over the steps (the
    launch sequence of a step is fixed: launch k of step i is the same kernel on the same shapes); (c) the result is refused
    (`rejected`) unless the probe step's own GPU time (events around the whole step) fits the replayed step + the markers' cost.
    The non-convolution floor of the step is measured in the run: replayed step time - event-timed convolution time.
    Returns (families, dominant-kernel record, totals, per-route table, rejected-or-None)."""
    from semantic_pyramid_for_image_generation_amd import ops
    runs = []
    host_ms = []
    eager_gpu_ms = []                         # GPU time of a whole probe step (first launch behind the blocker -> last launch), brackets included
    # one unrecorded eager step first: after replayed graphs the eager path has no VGG pyramid taken ahead (ModelWrapper._vgg_ahead), so
    # the FIRST eager step runs an extra pyramid pass - its launch list differs from every later step's (round-5 VERDICT, weak #3:
    # that difference made the per-launch median fall back to a single step, silently)
    step_fn()
    for _ in range(steps):
        ops.KERNEL_PROBE = []
        try:
            torch.cuda.synchronize()
            gpu_blocker(max(60.0, 3.0 * (host_ms[-1] if host_ms else 20.0)))
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            t0 = time.perf_counter()
            step_fn()
            host_ms.append((time.perf_counter() - t0) * 1e3)
            s1.record()
            torch.cuda.synchronize()
            eager_gpu_ms.append(s0.elapsed_time(s1))
            runs.append(ops.KERNEL_PROBE)
        finally:
            ops.KERNEL_PROBE = None
    # what a bracket ADDS to the kernel it encloses: K small identical kernels timed (a) by ONE pair of events around all of them and
    # (b) each inside its own bracket - the difference per kernel is the bracket's overhead with a kernel in it (an EMPTY bracket
    # measures more, ~4.6 us: with a kernel in flight the command processor handles the closing marker in its shadow; calibrated this
    # way the probe's convolution time agrees with rocprofv3's kernel time of the same box - profiles/README.md, round 4).  Subtracted
    # from every interval: 260 launches x ~2 us were 4 % of the convolution time the first version of this probe reported.
    from semantic_pyramid_for_image_generation_amd import _lib as _L
    cal_x = torch.zeros(4096, dtype=torch.float32, device="cuda")
    cal_y = torch.empty_like(cal_x)

    def tiny():
        _L.call("sp_act_fwd", ops.ptr(cal_x), ops.ptr(cal_y), cal_x.numel(), 0, _L.SP_F32, ops.stream())
    K = 200
    bracket_ms = None
    bracket_total_ms = None                   # what a bracket adds to the STREAM (its two markers), inside and outside the interval it measures
    for _ in range(3):
        torch.cuda.synchronize()
        gpu_blocker(10.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            tiny()
        e1.record()
        pairs = []
        for _ in range(K):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); tiny(); a1.record()
            pairs.append((a0, a1))
        torch.cuda.synchronize()
        est = (sum(a.elapsed_time(b) for a, b in pairs) - e0.elapsed_time(e1)) / K
        bracket_ms = est if bracket_ms is None else min(bracket_ms, est)
        tot = (pairs[0][0].elapsed_time(pairs[-1][1]) - e0.elapsed_time(e1)) / K
        bracket_total_ms = tot if bracket_total_ms is None else min(bracket_total_ms, tot)
    bracket_ms = max(bracket_ms, 0.0)
    bracket_total_ms = max(bracket_total_ms, bracket_ms)
    # steps whose launch list (flops, family, route, shape per launch) agrees with the LAST step's: with a fixed batch that is all of them
    sig = [tuple(a[2:] for a in r) for r in runs]
    runs = [r for r, g in zip(runs, sig) if g == sig[-1]]
    n = len(runs[0])
    if len(runs) < min(4, steps):
        raise RuntimeError("kernel_probe: only %d of %d eager steps issued the same launch list (%s launches) - a per-launch median over "
                           "them would not be what probe_method says" % (len(runs), steps, [len(g) for g in sig]))

    # the MEDIAN over the steps: the blocker already keeps the queue from running dry, and rocprofv3's per-kernel figures this is
    # compared with are means (the minimum of six picks every launch's best clock / cache state: 6 % under the profiler's sum)
    def med(v):
        v = sorted(v)
        return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])
    ms_min = [max(med([r[k][0].elapsed_time(r[k][1]) for r in runs]) - bracket_ms, 0.0) for k in range(n)]
    fam, routes = {}, {}
    dom_ms = dom_fl = 0.0
    dom_n = 0
    sn3 = {"dgrad": [0.0, 0.0, 0], "wgrad": [0.0, 0.0, 0]}        # the north-star's own sub-target: 3x3 spectral-norm layers of G and D, backward
    for ms, (_, _, fl, family, _mirror, route, shape, net) in zip(ms_min, runs[0]):
        dominant = is_dominant_route(route)   # the kernel the library CHOSE (sp_last_route), not a host-side mirror of its dispatch rule
        f = fam.setdefault(family, [0.0, 0.0, 0])
        f[0] += ms; f[1] += fl; f[2] += 1
        r = routes.setdefault((family, route), [0.0, 0.0, 0, None, 0.0])
        r[0] += ms; r[1] += fl; r[2] += 1
        if shape is not None and (r[3] is None or ms > r[4]):
            r[3], r[4] = shape, ms
        if dominant:
            dom_ms += ms; dom_fl += fl; dom_n += 1
        if net == "sn" and family in sn3 and shape is not None and shape[0] == 3:
            t = sn3[family]
            t[0] += ms; t[1] += fl; t[2] += 1
    families = {k: {"launches_per_step": v[2], "ms_per_step": round(v[0], 3), "gflop_per_step": round(v[1] / 1e9, 1),
                    "tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 1), "frac": round(v[1] / max(v[0], 1e-9) / 1e9 / peak, 4)}
                for k, v in sorted(fam.items())}
    table = [{"family": k[0], "route": k[1], "launches": v[2], "ms": round(v[0], 3), "gflop": round(v[1] / 1e9, 1),
              "tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 1),
              "slowest_launch": {"ksize_cin_cout_h_w_n": list(v[3]), "us": round(v[4] * 1e3, 1)} if v[3] is not None else None}
             for k, v in sorted(routes.items(), key=lambda kv: -kv[1][0])]
    tot_ms = sum(v[0] for v in fam.values())
    tot_fl = sum(v[1] for v in fam.values())
    bwd_ms = sum(v[0] for k, v in fam.items() if k != "fwd")
    bwd_fl = sum(v[1] for k, v in fam.items() if k != "fwd")
    nd = max(dom_n, 1)
    dom = {"kernel": DOMINANT_KERNEL, "launches_per_step": dom_n, "avg_launch_us": round(dom_ms / nd * 1e3, 2),
           "avg_algorithmic_gflop_per_launch": round(dom_fl / nd / 1e9, 3), "tflops": round(dom_fl / max(dom_ms, 1e-9) / 1e9, 2),
           "frac": round(dom_fl / max(dom_ms, 1e-9) / 1e9 / peak, 4), "ms_per_step": round(dom_ms, 3)}
    def sub(v):
        return {"launches": v[2], "ms": round(v[0], 3), "gflop": round(v[1] / 1e9, 1), "tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 1),
                "frac": round(v[1] / max(v[0], 1e-9) / 1e9 / peak, 4)}
    both = [sn3["dgrad"][i] + sn3["wgrad"][i] for i in range(3)]
    sn3_rec = dict(sub(both), input_gradient=sub(sn3["dgrad"]), weight_gradient=sub(sn3["wgrad"]),
                   scope="every 3x3 spectral-normalised convolution of G and D (the frozen VGG-16's input gradients are NOT in it): "
                         "input-gradient + weight-gradient launches of a step - BASELINE.json north_star's '>= 40 % MFMA utilisation on the "
                         "3x3 spectral-norm conv bwd'")
    eager_ms = sorted(eager_gpu_ms)[len(eager_gpu_ms) // 2]
    totals = {"tflops": round(tot_fl / max(tot_ms, 1e-9) / 1e9, 1), "ms_per_step": round(tot_ms, 3),
              "gflop_per_step": round(tot_fl / 1e9, 1), "launches": n,
              "backward_tflops": round(bwd_fl / max(bwd_ms, 1e-9) / 1e9, 1), "sn3x3_bwd": sn3_rec,
              "eager_probe_step_ms": round(eager_ms, 3), "bracket_stream_cost_us": round(bracket_total_ms * 1e3, 2), "probe_steps": len(runs),
              "method": "%d eager steps, each enqueued behind a %d+ ms stream blocker (host enqueue %.1f ms/step, never behind the GPU); "
                        "per-launch median over the steps, minus the %.2f us a bracket adds to the kernel inside it (calibrated on 200 small launches)"
                        % (len(runs), 60, sum(host_ms) / len(host_ms), bracket_ms * 1e3)}
    # Sanity of the event intervals, from THIS run only (round-4 VERDICT weak #5: the check leaned on a floor profiled on another box):
    # behind the blocker the probe step's GPU time is kernels + markers, so it must fit 1.05 x the replayed step + what the brackets
    # cost the stream; a probe step that took longer ran dry somewhere (host behind the GPU) and its intervals contain launch latency
    rejected = None
    if step_ms is not None:
        totals["nonconv_floor_ms"] = round(step_ms - tot_ms, 3)
        allowed = 1.05 * step_ms + n * bracket_total_ms
        if eager_ms > allowed or tot_ms > step_ms:
            rejected = ("probe step %.3f ms of GPU time > 1.05 x the measured step %.3f ms + %d brackets x %.2f us (or convolution time %.3f ms "
                        "> step): the event intervals contain something other than kernel time"
                        % (eager_ms, step_ms, n, bracket_total_ms * 1e3, tot_ms))
    return families, dom, totals, table, rejected



# linear layers + attention products of the step as the reference executes them (cf = 1: bmm 1.1 + addmm / mm 0.8
# GFLOP per image); not probed launch by launch (0.5 % of the step), added to the probed convolutions for the executed total
NONCONV_GFLOP_PER_IMAGE = {1: 1.9}


def roofline_fields(probe, peak, ips_per_gpu, batch, cf):
    """The probe's figures as they go into `roofline` (and into a sub-record): dominant kernel -> achieved / frac, all convolution
    launches -> conv_*, the north-star's 3x3 spectral-norm backward -> sn3x3_bwd, executed FLOPs -> step_*."""
    families, dom, totals, table, rejected = probe
    out = {}
    executed = totals["gflop_per_step"] / batch + NONCONV_GFLOP_PER_IMAGE.get(cf, 0.0)
    out["executed_gflop_per_image"] = round(executed, 2)
    out["executed_note"] = ("probed convolution launches on real channel counts (%.1f GFLOP per step / batch)%s; below the reference's count "
                            "where 1x1 residual convolutions run on the low-resolution side and the generator's feature mappings are reused"
                            % (totals["gflop_per_step"], " + %.1f linear / attention" % NONCONV_GFLOP_PER_IMAGE[cf] if cf in NONCONV_GFLOP_PER_IMAGE else ""))
    out["step_achieved"] = round(executed * ips_per_gpu / 1e3, 2)
    out["step_frac"] = round(executed * ips_per_gpu / 1e3 / peak, 4)
    if rejected is None:
        out.update({"achieved": dom["tflops"], "frac": dom["frac"],
                    "conv_achieved": totals["tflops"], "conv_frac": round(totals["tflops"] / peak, 4),
                    "backward_tflops": totals["backward_tflops"], "backward_frac": round(totals["backward_tflops"] / peak, 4),
                    "sn3x3_bwd": totals["sn3x3_bwd"]})
    else:
        out["probe_rejected"] = rejected              # frac stays null: an unsound number is not printed
    out.update({"conv_ms_per_step_eager": totals["ms_per_step"], "probe_method": totals["method"], "probe_steps": totals["probe_steps"],
                "nonconv_floor_ms": totals.get("nonconv_floor_ms"),
                "nonconv_floor_source": "measured in this run: replayed step time - event-timed convolution time",
                "eager_probe_step_ms": totals["eager_probe_step_ms"], "bracket_stream_cost_us": totals["bracket_stream_cost_us"],
                "families": families, "dominant_kernel": dom, "routes": table})
    return out


# ------------------------------------------------------------------------------------------------------------------
# one job = (channel factor, batch, dtype) on this rank
# ------------------------------------------------------------------------------------------------------------------
class Job:
    def __init__(self, cf, batch, dtype_name, dev, world, rank, use_graphs=True, device_masks=False):
        import semantic_pyramid_for_image_generation_amd as sp
        from semantic_pyramid_for_image_generation_amd import distributed, ops, params, synthetic
        self.batch, self.world = batch, world
        dtype = TORCH_DTYPE[dtype_name]
        ops.set_compute_dtype(dtype)
        torch.manual_seed(0)                                     # identical G/D init on every rank (default init, seed 0)
        G = sp.Generator(channels_factor=cf).to(dev)
        D = sp.Discriminator(channel_factor=cf).to(dev)
        V = sp.VGG16()
        V.load_state_dict(params.synth_state_dict(V.state_dict(), 2))      # kaiming-style weights: there is no pretrained file offline
        V.to(dev).eval()
        # torch.optim.Adam semantics / state (main.py:64-65), one multi-tensor launch per step (optim.py); SP_ADAM=torch keeps
        # torch's own foreach kernels for A/B runs
        adam = torch.optim.Adam if os.environ.get("SP_ADAM", "sempyr") == "torch" else sp.optim.Adam
        opt_g = adam(G.parameters(), lr=1e-5)
        opt_d = adam(D.parameters(), lr=1e-5)
        reducer = distributed.GradientReducer() if world > 1 else None
        self.mw = mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=opt_g, discriminator_optimizer=opt_d,
                                       save_data_path=None, gradient_reducer=reducer)
        G.train()
        D.train()
        images, labels, masks = synthetic.synthetic_batch(batch, 1234 + rank)
        images, labels, masks = images.to(dev), labels.to(dev), [m.to(dev) for m in masks]
        torch.manual_seed(100 + rank)                            # per-rank latent stream
        mask_gen = torch.Generator().manual_seed(7 + rank) if device_masks else None     # CPU generator: one host draw per batch, no device sync

        def fresh_masks():
            return synthetic.training_masks_device(batch, dev, mask_gen) if device_masks else masks

        def eager_step():
            # the resident batch is also the NEXT iteration's batch: its VGG pyramid rides in this iteration's pass over the fake images
            # (ModelWrapper.train_step: next_images_real), as it does for consecutive batches in ModelWrapper.train()
            return mw.train_step(images, labels, fresh_masks(), next_images_real=images)
        self.eager_step = eager_step
        self.step = eager_step
        self.launch_mode = "eager"
        if use_graphs:
            for _ in range(2):                                   # lazy state (packed VGG weights, kernel attributes) before the capture
                eager_step()
            try:
                mw.capture_graphs(images, labels, masks)

                def graphed_step():
                    return mw.train_step_graphed(None, None, fresh_masks() if device_masks else None)   # images / labels stay resident
                self.step = graphed_step
                self.launch_mode = "hipgraph"
            except Exception as exc:                             # capture is plumbing: fall back to eager launches, say so
                print("bench.py: HIP-graph capture failed (%s: %s); running eagerly" % (type(exc).__name__, exc), file=sys.stderr)
            if world > 1:
                # every rank must issue the SAME sequence of collectives: graph replay reduces the flat ranges after each graph,
                # eager launches hand groups over from inside the backward - one rank falling back alone would pair mismatched
                # buckets (or hang).  Agree on the minimum.
                import torch.distributed as dist
                ok = torch.tensor([1 if self.launch_mode == "hipgraph" else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok) == 0 and self.launch_mode == "hipgraph":
                    print("bench.py: rank %d captured, another rank did not - all ranks run eagerly" % rank, file=sys.stderr)
                    mw._graph_state = None
                    self.step, self.launch_mode = eager_step, "eager"

    def timed(self, steps, warmup):
        import torch.distributed as dist
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = self.step()
        torch.cuda.synchronize()
        elapsed_local = time.perf_counter() - t0
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        self.rank_elapsed = None
        if self.world > 1:
            mine = torch.tensor([elapsed_local], dtype=torch.float64, device="cuda")
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine)
            self.rank_elapsed = [float(x) for x in every]           # each rank's own time to its last step (before the closing barrier)
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        return elapsed, {k: float(v) for k, v in out.items() if k.startswith("loss")}

    def comm_stats(self, steps=5):
        """N > 1: a few extra steps with events around every collective (side stream) and every join (main stream):
        all-reduce time of the D and the G gradients and the part of it the main stream actually waited for."""
        red = self.mw.gradient_reducer
        if red is None or not red.active():
            return None
        red.timing = True
        try:
            for _ in range(steps):
                self.step()
            st = red.stats()
        finally:
            red.timing = False
        return st

    def close(self):
        self.mw = self.step = self.eager_step = None
        gc.collect()
        torch.cuda.empty_cache()


def sub_record(cf, batch, dtype_name, dev, steps, warmup, use_graphs, probe=False):
    """probe: also the per-launch convolution probe of this configuration (families / routes / executed FLOPs), as in the headline."""
    job = Job(cf, batch, dtype_name, dev, 1, 0, use_graphs)
    elapsed, _ = job.timed(steps, warmup)
    mode = job.launch_mode
    ips = batch * steps / elapsed
    gf = GFLOP_PER_IMAGE.get(cf)
    peak = PEAK_TFLOPS[dtype_name]
    rec = {"dtype": dtype_name, "batch": batch, "channel_factor": cf, "value": round(ips, 2), "unit": "images/sec",
           "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup, "launch": mode}
    if gf:
        rec["step_tflops_reference_flops"] = round(gf * ips / 1e3, 1)
        rec["step_frac_reference_flops"] = round(gf * ips / 1e3 / peak, 4)
        rec["reference_gflop_per_image"] = gf
    if probe:
        try:
            pr = kernel_probe(job.eager_step, peak, steps=4, step_ms=elapsed / steps * 1e3)
            rf = roofline_fields(pr, peak, ips, batch, cf)
            rf.pop("probe_method", None)
            rec["roofline"] = rf
        except Exception as exc:
            rec["roofline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    job.close()
    return rec


def bf16_parity_record(dev, tag="step_cf1_b2_seed0", dtype_name="bf16"):
    """The benchmarked (bf16) mode against the committed reference goldens of the two-iteration loop (tests/golden/<tag>: loss
    scalars and 4096 generator-pixel samples recorded from the unmodified reference, tests/golden/make_golden.py): the MEASURED
    errors of this very build, so that the throughput figure and "matches the reference" are statements about one program.
    tests/test_gpu_step.py::test_train_step_bf16_restated_tolerance asserts 2x these figures."""
    import json as _json
    import numpy as np
    import semantic_pyramid_for_image_generation_amd as sp
    from semantic_pyramid_for_image_generation_amd import ops, params
    gold = os.path.join(ROOT, "tests", "golden")
    if gold not in sys.path:
        sys.path.insert(0, gold)
    import make_golden                                           # committed generator of the golden batches (data only)
    meta = _json.load(open(os.path.join(gold, tag + ".json")))
    arr = dict(np.load(os.path.join(gold, tag + ".npz")))
    ops.set_compute_dtype(TORCH_DTYPE[dtype_name])
    G, D, V = sp.Generator(channels_factor=meta["cf"]), sp.Discriminator(channel_factor=meta["cf"]), sp.VGG16()
    for net, seed in ((G, meta["seed"]), (D, meta["seed"] + 1), (V, meta["seed"] + 2)):
        net.load_state_dict(params.synth_state_dict(net.state_dict(), seed))
    G.to(dev).train(); D.to(dev).train(); V.to(dev).eval()
    mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=torch.optim.Adam(G.parameters(), lr=meta["lr"]),
                         discriminator_optimizer=torch.optim.Adam(D.parameters(), lr=meta["lr"]), save_data_path=None)
    noise = torch.from_numpy(arr["noise"]).to(dev)
    names = ("loss_discriminator_real", "loss_discriminator_fake", "loss_generator", "loss_generator_semantic_reconstruction",
             "loss_generator_diversity")
    g = torch.Generator().manual_seed(977)
    idx = torch.randint(0, meta["batch_size"] * 3 * 256 * 256, (4096,), generator=g)
    rec = {"loss_rel": [], "pixel_max": [], "pixel_rms": []}
    for it, (images, labels, masks) in enumerate(make_golden.golden_batches(meta["batch_size"], meta["seed"])):
        out = mw.train_step(images.to(dev), labels.to(dev), [m.to(dev) for m in masks], noise_d=noise[2 * it], noise_g=noise[2 * it + 1])
        rec["loss_rel"].append(max(abs(float(out[n]) - meta[n][it]) / max(abs(meta[n][it]), 2e-2) for n in names))
        fake = out["images_fake"].float().cpu().contiguous().flatten()[idx].numpy()
        ref = arr["fake_samples"][2 * it + 1]
        rec["pixel_max"].append(float(np.abs(fake - ref).max()))
        rec["pixel_rms"].append(float(np.sqrt(np.mean((fake - ref) ** 2))))
    del mw
    gc.collect()
    torch.cuda.empty_cache()
    # what 16-bit storage costs by itself: the CPU oracle with every layer output, gradient and normalised weight rounded to the storage
    # type (oracle.set_storage - no kernel involved) against the same goldens; the tests bound the GPU modes by 2x these figures
    model = None
    try:
        tests_dir = os.path.join(ROOT, "tests")
        if tests_dir not in sys.path:
            sys.path.insert(0, tests_dir)
        import golden_util
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)))
        m = golden_util.storage_noise_model(tag, TORCH_DTYPE[dtype_name], 65536.0 if dtype_name == "fp16" else 1.0)
        model = {"worst_loss_rel_err": round(max(m["loss_rel"]), 6), "worst_pixel_abs_err": round(max(m["pixel_max"]), 5),
                 "pixel_rms_err": round(max(m["pixel_rms"]), 5)}
    except Exception as exc:
        model = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return {"against": "reference goldens tests/golden/%s (cf=%s, batch %d, 2 iterations of the reference's own loop)" % (tag, meta["cf"], meta["batch_size"]),
            "worst_loss_rel_err": round(max(rec["loss_rel"]), 6), "worst_pixel_abs_err": round(max(rec["pixel_max"]), 5),
            "pixel_rms_err": round(max(rec["pixel_rms"]), 5), "fp32_mode_bound": 1e-3, "oracle_storage_noise_model": model,
            "note": "%s storage + %s MFMA + fp32 accumulate vs the fp32 reference; the fp32 parity mode meets 1e-3 (parity_mode record)" % (dtype_name, dtype_name)}


def flush_c_stdio():
    """RCCL writes its banner with C stdio, which is fully buffered on a pipe and would otherwise surface after the JSON line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


# ------------------------------------------------------------------------------------------------------------------
# box telemetry (round-5 VERDICT, weak #9: pool boxes differ by +-5 %; the line must say which kind of box it ran on)
# ------------------------------------------------------------------------------------------------------------------
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _gpu_sysfs_dir(dev_index=0):
    """/sys/bus/pci/devices/<domain:bus:device.0> of torch's device (plain file reads: nothing is executed, no GPU call is made)."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        return d if os.path.isdir(d) else None
    except Exception:
        return None


def _hwmon_dir(sysdir):
    try:
        base = os.path.join(sysdir, "hwmon")
        for h in sorted(os.listdir(base)):
            return os.path.join(base, h)
    except (OSError, TypeError):
        pass
    return None


def _dpm_levels(text):
    """pp_dpm_sclk / pp_dpm_mclk: '0: 132Mhz\n1: 2400Mhz *' -> (levels in MHz, current level in MHz)."""
    levels, cur = [], None
    for ln in (text or "").splitlines():
        m = re.search(r"(\d+)\s*mhz", ln.lower())
        if m:
            levels.append(int(m.group(1)))
            if "*" in ln:
                cur = int(m.group(1))
    return levels, cur


def gpu_sample(sysdir):
    """One reading of the shader / memory clock (MHz) and the board power (W) from sysfs; missing files give None."""
    out = {"sclk_mhz": None, "mclk_mhz": None, "power_w": None}
    if sysdir is None:
        return out
    hw = _hwmon_dir(sysdir)
    if hw is not None:
        f = _read(os.path.join(hw, "freq1_input"))
        if f and f.isdigit():
            out["sclk_mhz"] = int(f) // 1000000
        f = _read(os.path.join(hw, "freq2_input"))
        if f and f.isdigit():
            out["mclk_mhz"] = int(f) // 1000000
        for name in ("power1_average", "power1_input"):
            f = _read(os.path.join(hw, name))
            if f and f.isdigit():
                out["power_w"] = round(int(f) / 1e6, 1)
                break
    if out["sclk_mhz"] is None:
        out["sclk_mhz"] = _dpm_levels(_read(os.path.join(sysdir, "pp_dpm_sclk")))[1]
    if out["mclk_mhz"] is None:
        out["mclk_mhz"] = _dpm_levels(_read(os.path.join(sysdir, "pp_dpm_mclk")))[1]
    return out


def box_static(sysdir):
    """What does not change during the run: clock ceilings, the power cap, the performance level the box is pinned to."""
    out = {}
    if sysdir is None:
        return out
    lv, _ = _dpm_levels(_read(os.path.join(sysdir, "pp_dpm_sclk")))
    if lv:
        out["sclk_max_mhz"] = max(lv)
    lv, _ = _dpm_levels(_read(os.path.join(sysdir, "pp_dpm_mclk")))
    if lv:
        out["mclk_max_mhz"] = max(lv)
    hw = _hwmon_dir(sysdir)
    if hw is not None:
        f = _read(os.path.join(hw, "power1_cap"))
        if f and f.isdigit():
            out["power_cap_w"] = round(int(f) / 1e6)
    f = _read(os.path.join(sysdir, "power_dpm_force_performance_level"))
    if f:
        out["perf_level"] = f
    return out


def rocm_smi_snapshot():
    """Fallback where sysfs is not readable: one `rocm-smi` call as a child process, made BEFORE this process touches the GPU."""
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=30)
        js = json.loads(r.stdout)
        card = js[sorted(js)[0]]
        out = {}
        for k, v in card.items():
            kl = k.lower()
            m = re.search(r"\((\d+)mhz\)", str(v).lower())
            if kl.startswith("sclk") and m:
                out["sclk_mhz_idle"] = int(m.group(1))
            elif kl.startswith("mclk") and m:
                out["mclk_mhz_idle"] = int(m.group(1))
            elif "max graphics package power" in kl:
                out["power_cap_w"] = round(float(v))
            elif "graphics package power" in kl or "average graphics package power" in kl:
                out["power_w_idle"] = round(float(v), 1)
        return out
    except Exception:
        return {}


class ClockSampler:
    """Reads the clocks and the power every 50 ms on a host thread while a timed window runs: what the box sustained UNDER LOAD."""

    def __init__(self, sysdir):
        self.sysdir, self.samples, self._stop, self._th = sysdir, [], False, None

    def __enter__(self):
        if self.sysdir is not None:
            import threading

            def loop():
                while not self._stop:
                    self.samples.append(gpu_sample(self.sysdir))
                    time.sleep(0.05)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=2)

    def summary(self):
        def med(key):
            v = sorted(x[key] for x in self.samples if x.get(key) is not None)
            return v[len(v) // 2] if v else None
        return {"sclk_mhz_load": med("sclk_mhz"), "mclk_mhz_load": med("mclk_mhz"), "power_w_load": med("power_w"), "samples": len(self.samples)}


# ------------------------------------------------------------------------------------------------------------------
# the compact line (round-5 VERDICT, next #1: a 31.6 KB line outgrew the driver's capture and the headline was lost)
# ------------------------------------------------------------------------------------------------------------------
DETAIL_FILE = "bench_detail.json"
COMPACT_LIMIT = 4096


def _pick(d, *keys):
    """d[k0][k1]... or None."""
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(detail: dict) -> dict:
    """The record printed as the LAST stdout line: the contract's fields + one number per sub-record; everything else stays in
    `detail` (bench_detail.json, stderr).  Pure function of the full record (tests/test_bench_line.py holds it under COMPACT_LIMIT)."""
    line = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data")}
    cfg = detail.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "global_batch", "parallelism", "launch") if k in cfg}
    if cfg.get("collectives"):
        line["config"]["collectives"] = cfg["collectives"]
    rf = detail.get("roofline") or {}
    dom = rf.get("dominant_kernel") or {}
    line["roofline"] = {
        "bound": rf.get("bound"), "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": rf.get("frac"),
        "traffic": rf.get("traffic"), "step_frac": rf.get("step_frac"), "step_frac_reference_flops": rf.get("step_frac_reference_flops"),
        "conv_frac": rf.get("conv_frac"), "backward_frac": rf.get("backward_frac"),
        "sn3x3_bwd_frac": _pick(rf, "sn3x3_bwd", "frac"), "sn3x3_dgrad_frac": _pick(rf, "sn3x3_bwd", "input_gradient", "frac"),
        "sn3x3_wgrad_frac": _pick(rf, "sn3x3_bwd", "weight_gradient", "frac"),
        "nonconv_floor_ms": rf.get("nonconv_floor_ms"), "probe_steps": rf.get("probe_steps"),
        "dominant_kernel": {"name": "conv3x3_pp_kernel<bf16,2,...,FW=2>" if detail.get("dtype") != "f32" else "conv3x3_tall_kernel<f32,2,8>",
                            "launches": dom.get("launches_per_step"), "avg_us": dom.get("avg_launch_us"),
                            "gflop": dom.get("avg_algorithmic_gflop_per_launch")},
    }
    if rf.get("probe_rejected"):
        line["roofline"]["probe_rejected"] = str(rf["probe_rejected"])[:160]
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "cpu_model": cb.get("cpu_model"), "batch20": _pick(cb, "batch20", "value"),
                                "sample": "oracle/sempyr_oracle.py (torch fp32): %s batch-2 steps -> value, %s batch-20 steps -> batch20; threads "
                                          "capped at %s of %s" % (cb.get("steps_b2"), _pick(cb, "batch20", "steps"), cb.get("cores"),
                                                                  cb.get("host_cores_available"))}
    sus = detail.get("sustained")
    if isinstance(sus, dict):
        line["sustained"] = {k: sus.get(k) for k in ("value", "ms_per_step", "steps")}
    for key in ("parity_mode", "batch32", "fp16", "deterministic", "channel_factor2", "channel_factor0.5"):
        rec = detail.get(key)
        if isinstance(rec, dict):
            line[key] = rec.get("value") if "error" not in rec else None
    for key in ("channel_factor2", "channel_factor0.5"):
        v = _pick(detail, key, "roofline", "sn3x3_bwd", "frac")
        if v is not None:
            line[key + "_sn3x3_bwd_frac"] = v
    par = {}
    for key, src in (("fp32_b20_pixels", ("parity_b20", "worst_pixel_abs_err")), ("fp32_b20_losses", ("parity_b20", "worst_loss_rel_err")),
                     ("bf16_pixels_rms", ("bf16_parity", "pixel_rms_err")), ("bf16_pixels_worst", ("bf16_parity", "worst_pixel_abs_err")),
                     ("bf16_losses", ("bf16_parity", "worst_loss_rel_err")),
                     ("fp16_pixels_rms", ("fp16", "parity", "pixel_rms_err")), ("fp16_pixels_worst", ("fp16", "parity", "worst_pixel_abs_err")),
                     ("fp16_losses", ("fp16", "parity", "worst_loss_rel_err"))):
        v = _pick(detail, *src)
        if v is not None:
            par[key] = v
    if par:
        line["parity"] = par
    mg = detail.get("multi_gpu")
    if isinstance(mg, dict):
        line["multi_gpu"] = {"per_rank": mg.get("per_rank_images_per_sec"), "allreduce_ms_d": mg.get("allreduce_ms_d"),
                             "allreduce_ms_g": mg.get("allreduce_ms_g"), "exposed_ms": mg.get("exposed_ms"),
                             "grad_bytes_d": mg.get("grad_bytes_d"), "grad_bytes_g": mg.get("grad_bytes_g")}
    if detail.get("box"):
        line["box"] = detail["box"]
    line["detail"] = DETAIL_FILE
    return line


def emit(detail: dict) -> None:
    """Full record -> bench_detail.json + stderr; compact record -> the last stdout line."""
    line = compact_line(detail)
    text = json.dumps(line)
    if len(text) > COMPACT_LIMIT:                    # never again a line the driver cannot hold: drop optional groups, largest first
        for key in ("parity", "box", "multi_gpu"):
            line.pop(key, None)
            text = json.dumps(line)
            if len(text) <= COMPACT_LIMIT:
                break
    full = json.dumps(detail)
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
            f.write(full + "\n")
    except OSError as exc:
        print("bench.py: could not write %s (%s)" % (DETAIL_FILE, exc), file=sys.stderr)
    print(full, file=sys.stderr, flush=True)
    flush_c_stdio()
    print(text, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))             # nothing above touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # BENCH_ONE_DEVICE=1 + BENCH_DIST_BACKEND=gloo: every rank on cuda:0, gloo between them - the dry run of the N > 1 code path on a
    # one-GPU box (tests/test_gpu_two_ranks.py; RCCL refuses two ranks on one device).  Never a measurement.
    one_device = os.environ.get("BENCH_ONE_DEVICE", "0") == "1"
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    smi = rocm_smi_snapshot() if (rank == 0 and os.environ.get("BENCH_NO_SMI", "0") != "1") else {}    # a child process, BEFORE the first GPU call
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()                               # creates the communicator now: RCCL prints its version banner here ...
        flush_c_stdio()                              # ... through C stdio - push it out BEFORE the JSON line, not at exit
    sysdir = _gpu_sysfs_dir(local_rank)
    box = dict(smi, **box_static(sysdir))
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        box.update({"gpu": pr.name, "cus": pr.multi_processor_count})
    except Exception:
        pass
    idle = gpu_sample(sysdir)
    if idle["sclk_mhz"] is not None:
        box["sclk_mhz_idle"] = idle["sclk_mhz"]

    cf = args.channel_factor if args.channel_factor != int(args.channel_factor) else int(args.channel_factor)
    if args.fp8:
        from semantic_pyramid_for_image_generation_amd import ops as _ops0
        _ops0.set_vgg_fp8(1)
    if args.deterministic:
        from semantic_pyramid_for_image_generation_amd import ops as _ops1
        _ops1.set_tuning(_ops1.TUNE_DETERMINISTIC, 1)
    job = Job(cf, args.batch, args.dtype, dev, world, rank, not args.no_graphs, args.device_masks)
    # N = 1: the >= 6 s window first (power / clock steady state, sampled from sysfs while it runs), then - on the same job, with no
    # idle gap in between - the contract's W warm-up + K timed steps, which are the headline
    sustained = None
    if world == 1 and not args.no_sub_records:
        t_est, _ = job.timed(5, max(args.warmup, 3))
        n_sus = max(args.steps, int(6.0 / max(t_est / 5, 1e-4)))
        with ClockSampler(sysdir) as cs:
            t_sus, _ = job.timed(n_sus, 0)
        box.update(cs.summary())
        sustained = {"value": round(args.batch * n_sus / t_sus, 2), "unit": "images/sec", "ms_per_step": round(t_sus / n_sus * 1e3, 3),
                     "steps": n_sus, "note": ">= 6 s of back-to-back steps (power / clock steady state) right before the headline's K steps"}
    elapsed, losses = job.timed(args.steps, args.warmup)
    # the probe runs extra training steps: with world > 1 they contain collectives, so EVERY rank takes them
    probe = None
    probe_error = None
    peak = PEAK_TFLOPS[args.dtype]
    if not args.no_kernel_probe:
        try:
            probe = kernel_probe(job.eager_step, peak, steps=5, step_ms=elapsed / args.steps * 1e3)   # the probe brackets individual launches: eager steps
        except RuntimeError as exc:          # (raised after the probe's steps have run: the ranks are still in step with each other)
            probe_error = str(exc)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    launch_mode = job.launch_mode
    comm = job.comm_stats() if world > 1 else None
    grad_bytes = {k: 4 * int(bk.flat.numel()) for k, bk in job.mw._banks.items() if bk.flat is not None}
    rank_elapsed = job.rank_elapsed
    job.close()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        ips = args.batch * world * args.steps / elapsed
        gf = GFLOP_PER_IMAGE.get(cf)
        achieved = (gf * ips / world / 1e3) if gf else None           # TFLOP/s per GPU, algorithmic
        line = {
            "metric": "images/sec full G+D train step, 256x256, bs/GPU=%d" % args.batch,
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "Semantic-Pyramid GAN D+G step, channel_factor=%g, 256x256x3, batch %d/GPU, Adam lr 1e-5, "
                                   "random-init G/D, kaiming-init frozen VGG-16" % (cf, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "launch": launch_mode, "collectives": (backend + (", every rank on one device: a dry run, not a measurement" if one_device else "")) if world > 1 else None,
                       "deterministic": bool(args.deterministic or args.dtype == "f32"), "masks": "fresh on-device batch per step" if args.device_masks else "one resident batch",
                       "losses_last_step": losses},
            "roofline": {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                         "basis": "achieved / frac: the DOMINANT KERNEL alone, as the bench contract defines roofline (dominant_kernel: algorithmic "
                                  "FLOPs of its launches / their event-timed duration) - NOT the step's figure; conv_frac: FLOP-weighted over EVERY "
                                  "convolution launch of a step (forward + input gradient + weight gradient; this was `frac` up to round 4); "
                                  "step_frac: the WHOLE step, executed FLOPs per image x images/s per GPU - the figure to read the headline against",
                         "step_achieved_reference_flops": round(achieved, 2) if achieved else None,
                         "step_frac_reference_flops": round(achieved / peak, 4) if achieved else None,
                         "reference_gflop_per_image": gf,
                         "sustained_mfma_peak": {"value": 2010.0, "unit": "TFLOP/s",
                                                 "note": "recorded, not measured in this run: a registers-only v_mfma_f32_16x16x32_bf16 stream "
                                                         "on every CU with random operands (profiles/README.md) - the "
                                                         "power budget gives the matrix pipe 80 % of `peak`; frac stays priced on `peak`"}},
            "box": box,
        }
        if sustained is not None:
            line["sustained"] = sustained
        if probe_error is not None:
            line["roofline"]["probe_rejected"] = probe_error
        if world > 1:
            # what a bad scaling curve needs to be diagnosed: the collectives' own time, what of it was NOT hidden, each rank's pace
            line["multi_gpu"] = {
                "per_rank_images_per_sec": [round(args.batch * args.steps / t, 2) for t in (rank_elapsed or [])],
                "allreduce_ms_d": comm["d"]["allreduce_ms"] if comm and "d" in comm else None,
                "allreduce_ms_g": comm["g"]["allreduce_ms"] if comm and "g" in comm else None,
                "exposed_ms": round(sum(v["exposed_ms"] for v in comm.values()), 4) if comm else None,
                "exposed_ms_d": comm["d"]["exposed_ms"] if comm and "d" in comm else None,
                "exposed_ms_g": comm["g"]["exposed_ms"] if comm and "g" in comm else None,
                "note": "rank 0's events over 5 extra steps: allreduce_ms = first collective start -> last end on the side stream; "
                        "exposed_ms = main-stream wait at the joins (D's join sits behind the generator forward, G's before Adam(G))"}
            # what to EXPECT, so that a bad curve can be read from the line alone: bytes on the wire and the time a ring all-reduce of them
            # takes at the xGMI figures of MI355X_MICROARCH.md (7 links x ~153 GB/s per GPU; a ring is bound by ONE link direction per hop:
            # t = 2 (N - 1) / N x bytes / link_bw); D's reduction hides behind the generator forward (~3 ms), G's is exposed in full
            mg = line["multi_gpu"]
            for key, nb in (("d", grad_bytes.get("d")), ("g", grad_bytes.get("g"))):
                if nb:
                    mg["grad_bytes_" + key] = nb
                    mg["expected_ring_ms_" + key] = round(2.0 * (world - 1) / world * nb / 153e9 * 1e3, 3)
                    ms_meas = mg.get("allreduce_ms_" + key)
                    if ms_meas:
                        mg["measured_algbw_gbps_" + key] = round(nb / (ms_meas * 1e-3) / 1e9, 1)
            mg["expected_exposed_ms_g"] = mg.get("expected_ring_ms_g")
        if probe is not None:
            line["roofline"].update(roofline_fields(probe, peak, ips / world, args.batch, cf))
            traffic, prov = recorded_traffic((DOMINANT_KERNEL_SYMBOL, "conv3x3_pp_kernel<bf16, 8", "conv3x3_tall_kernel<bf16, 2, 8>"))
            line["roofline"]["traffic"] = traffic
            line["roofline"]["traffic_source"] = prov
        subs = world == 1 and not args.no_sub_records
        graphs = not args.no_graphs
        if subs and args.dtype == "bf16":
            # the 16-bit storage mode that is eight times closer to the fp32 reference at ~99 % of the bf16 speed, FIRST after the headline
            # (BASELINE.json config 5's activations; its fp8 half is retired from the default line - see DESIGN.md "config 5": the e4m3
            # slice of the frozen pyramid is slower than fp16 alone and fails the gradient rule of tests/test_gpu_fp8.py; --fp8 still runs it)
            try:
                line["fp16"] = sub_record(cf, args.batch, "fp16", dev, 20, 5, graphs)
                line["fp16"]["vs_bf16_headline"] = round(line["fp16"]["value"] / ips, 4)
                if cf == 1:
                    line["fp16"]["parity"] = bf16_parity_record(dev, dtype_name="fp16")
                line["fp16"]["note"] = ("bench.py --dtype fp16: the same kernels on IEEE half storage / v_mfma_f32_16x16x32_f16, fp32 accumulate, dynamic "
                                        "loss scale; `parity` = measured error of that mode against the reference goldens (bf16's is `bf16_parity`)")
            except Exception as exc:
                line["fp16"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if subs and args.dtype == "bf16":
            line["parity_mode"] = sub_record(cf, args.batch, "f32", dev, 8, 3, graphs)
            line["parity_mode"]["note"] = ("fp32 storage, exact-fp32 MFMA, ordered (deterministic) reductions: the mode held to 1e-3 "
                                           "on pixels and losses against the reference goldens")
        if subs and args.batch != 32:
            line["batch32"] = sub_record(cf, 32, args.dtype, dev, 15, 5, graphs)
            line["batch32"]["note"] = "BASELINE.json config 2: one MI355X, batch 32"
        if subs and args.dtype != "f32" and not args.deterministic:
            # the same step with every reduction in a fixed order (bit-identical from run to run, tests/test_gpu_step.py): its price
            from semantic_pyramid_for_image_generation_amd import ops as _ops2
            _ops2.set_tuning(_ops2.TUNE_DETERMINISTIC, 1)
            try:
                line["deterministic"] = sub_record(cf, args.batch, args.dtype, dev, 15, 5, graphs)
                line["deterministic"]["vs_headline"] = round(line["deterministic"]["value"] / ips, 4)
                line["deterministic"]["note"] = "SP_TUNE_DETERMINISTIC=1 (bench.py --deterministic): no fp32 atomics anywhere in the step"
            except Exception as exc:
                line["deterministic"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            finally:
                _ops2.set_tuning(_ops2.TUNE_DETERMINISTIC, -1)
        if subs and args.dtype == "bf16" and cf == 1:
            try:
                line["bf16_parity"] = bf16_parity_record(dev)
            except Exception as exc:                             # a reported figure, not the thing measured: never sink the line
                line["bf16_parity"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if subs and args.dtype == "bf16" and cf == 1:
            # BASELINE.json config 4 on one GPU's share: the narrow (channel_factor 2: 512 // 2 channels) and the wide (0.5) networks at
            # the metric's batch, each with its own launch probe (families, routes, the 3x3 spectral-norm backward, executed FLOPs)
            for key, other in (("channel_factor2", 2), ("channel_factor0.5", 0.5)):
                try:
                    line[key] = sub_record(other, args.batch, args.dtype, dev, 10, 4, graphs, probe=not args.no_kernel_probe)
                    line[key]["note"] = "BASELINE.json config 4 (/root/reference/models.py:34-48,117-128: the factor divides), one GPU, batch %d" % args.batch
                except Exception as exc:
                    line[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cf)
            ref20 = line["cpu_baseline"].pop("_ref20", None)
            if ref20 is not None:
                try:
                    line["parity_b20"] = parity_b20_record(dev, ref20, cf)
                except Exception as exc:                         # a reported figure: never sink the line
                    line["parity_b20"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
