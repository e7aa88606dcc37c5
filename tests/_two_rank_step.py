"""The REAL training step with two data-parallel ranks on ONE GPU (round-4 VERDICT, "Next round" #2; replaces
/root/reference/main.py:91-94's nn.DataParallel).

    python tests/_two_rank_step.py launch <out.json>        started by tests/conftest.py from a process that has not touched the GPU
    python tests/_two_rank_step.py rank <rank> <world> <port> <out prefix>

Both ranks sit on cuda:0 and talk through gloo (RCCL refuses two ranks on one device; gloo all-reduces CUDA tensors through
the host), so everything that is NOT the wire runs exactly as on a multi-GPU node: ModelWrapper + distributed.GradientReducer,
the flat fp32 gradient buffers, the side stream, the group hooks of the eager backward, the bucketed ranges behind the
replayed graphs.  Every rank owns its own shard (other images / labels / masks / latents, identical parameters) for two
iterations, once with eager launches and once with iteration 1 replayed from captured HIP graphs, fp32 parity mode.

Expectation = the CPU oracle (test infrastructure) run on the same shard with optimizers whose step() first AVERAGES the
gradients over the ranks (plain dist.all_reduce of CPU tensors - not the reducer under test): "average of the shards'
gradients -> Adam", the contract of SURVEY.md section 8e.  Checked per rank: the five losses of every iteration, the reduced
D and G gradients that reach Adam (Adam's update is invariant to a constant gradient scale, so a missing 1/N would not show
in the parameters), every parameter after two iterations; across ranks: parameters, spectral-norm u / v and the reduced
gradients are bit-identical; eager and replayed runs agree bit for bit."""
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CF, BATCH, LR, ITERS = 8, 2, 1e-4, 2
LOSSES = (("loss_discriminator_real", "loss_d_real"), ("loss_discriminator_fake", "loss_d_fake"), ("loss_generator", "loss_g"),
          ("loss_generator_semantic_reconstruction", "loss_rec"), ("loss_generator_diversity", "loss_div"))


def shard_batches(rank):
    """ITERS (images, labels, masks, noise_d, noise_g) of this rank's shard, on the CPU."""
    import torch
    from semantic_pyramid_for_image_generation_amd import synthetic
    out = []
    for it in range(ITERS):
        images, labels, masks = synthetic.synthetic_batch(BATCH, 500 + 10 * it + rank)
        g = torch.Generator().manual_seed(900 + 10 * it + rank)
        out.append((images, labels, masks, torch.randn(BATCH, 128, generator=g), torch.randn(BATCH, 128, generator=g)))
    return out


class AveragingAdam:
    """The oracle side's optimizer: step() = all-reduce(mean) of the gradients over the ranks, then torch.optim.Adam."""

    def __init__(self, params, lr, world):
        import torch
        self.params, self.world = list(params), world
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.reduced = []                      # the averaged gradients of every step, in parameter order

    def step(self):
        import torch
        import torch.distributed as dist
        grads = [p.grad for p in self.params]
        flat = torch._utils._flatten_dense_tensors(grads)
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= self.world
        for g, r in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(r)
        self.reduced.append([g.detach().clone() for g in grads])
        self.opt.step()


def oracle_run(rank, world):
    import torch
    from oracle import sempyr_oracle as O
    from semantic_pyramid_for_image_generation_amd import params
    G = O.make_state(params.synth_state_dict(O.layout_template(O.generator_layout(CF)), 0))
    D = O.make_state(params.synth_state_dict(O.layout_template(O.discriminator_layout(CF)), 1))
    V = O.make_state(params.synth_state_dict(O.layout_template(O.vgg16_layout()), 2), frozen=True)
    og, od = AveragingAdam(O.trainable(G), LR, world), AveragingAdam(O.trainable(D), LR, world)
    losses = []
    for images, labels, masks, nd, ng in shard_batches(rank):
        out = O.train_step(G, D, V, og, od, images, labels, masks, nd, ng, skip_dead_d_wgrad=True)
        losses.append({r: float(out[r]) for _, r in LOSSES})
    return G, D, og.reduced, od.reduced, losses


def gpu_run(rank, graphed):
    """Two iterations of ModelWrapper on this rank's shard.  graphed: iteration 0 eager, then capture, iteration 1 replayed."""
    import torch
    import semantic_pyramid_for_image_generation_amd as sp
    from semantic_pyramid_for_image_generation_amd import distributed, ops, params
    from oracle import sempyr_oracle as O
    ops.set_compute_dtype(torch.float32)
    G, D, V = sp.Generator(channels_factor=CF), sp.Discriminator(channel_factor=CF), sp.VGG16()
    G.load_state_dict(params.synth_state_dict(O.layout_template(O.generator_layout(CF)), 0))
    D.load_state_dict(params.synth_state_dict(O.layout_template(O.discriminator_layout(CF)), 1))
    V.load_state_dict(params.synth_state_dict(O.layout_template(O.vgg16_layout()), 2))
    G.cuda().train(); D.cuda().train(); V.cuda().eval()
    opt_g, opt_d = torch.optim.Adam(G.parameters(), lr=LR), torch.optim.Adam(D.parameters(), lr=LR)
    red = distributed.GradientReducer(bucket_bytes=1 << 18)          # small buckets: several collectives per network
    mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=opt_g, discriminator_optimizer=opt_d, save_data_path=None,
                         gradient_reducer=red)
    assert mw._reducer_active()
    reduced = {"d": [], "g": []}
    for key, opt, net in (("d", opt_d, D), ("g", opt_g, G)):
        def spy(orig=opt.step, key=key, net=net):
            reduced[key].append([p.grad.detach().float().cpu().clone() for p in net.parameters()])
            return orig()
        opt.step = spy
    losses = []
    for it, (images, labels, masks, nd, ng) in enumerate(shard_batches(rank)):
        images, labels, masks, nd, ng = images.cuda(), labels.cuda(), [m.cuda() for m in masks], nd.cuda(), ng.cuda()
        if graphed and it == ITERS - 1:
            mw.capture_graphs(images, labels, masks)
            out = mw.train_step_graphed(images, labels, masks, noise_d=nd, noise_g=ng)
        else:
            # the last iteration announces its own batch as its successor, as the capture does (capture_graphs: images_next): the frozen
            # VGG-16 then takes [fake | next real] in one pass in both launch modes, and the two runs must agree bit for bit
            out = mw.train_step(images, labels, masks, noise_d=nd, noise_g=ng, next_images_real=images if it == ITERS - 1 else None)
        losses.append({a: float(out[a]) for a, _ in LOSSES})
    torch.cuda.synchronize()
    return G, D, reduced, losses


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().float().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def compare(rank, world, mode, gpu, oracle):
    """-> (list of failures, record of the worst measured errors)."""
    import torch
    from oracle import sempyr_oracle as O
    G, D, reduced, losses = gpu
    oG, oD, o_red_g, o_red_d, o_losses = oracle
    bad, rec = [], {}
    worst = 0.0
    for it in range(ITERS):
        for a, r in LOSSES:
            got, want = losses[it][a], o_losses[it][r]
            err = abs(got - want) / max(abs(want), 1e-3)
            worst = max(worst, err)
            if err > 1e-3:
                bad.append("%s rank %d iteration %d: %s = %.7g, oracle %.7g" % (mode, rank, it, a, got, want))
    rec["worst_loss_rel_err"] = worst
    # the gradients Adam saw: per-tensor norm and element-wise, 5e-3 at iteration 0 / 1e-2 behind an Adam step (tests/test_gpu_step.py)
    for key, net, o_red in (("d", D, o_red_d), ("g", G, o_red_g)):
        worst_n = worst_e = 0.0
        for it in range(ITERS):
            rtol = 5e-3 if it == 0 else 1e-2
            ref = o_red[it]
            top = max(float(r.norm()) for r in ref)
            for (name, _), g, r in zip(net.named_parameters(), reduced[key][it], ref):
                # tests/test_gpu_step.py's model: |norm - ref| <= rtol * ref + 1e-5 * (largest norm of the network) - the biases in
                # front of a BatchNorm have a true gradient of zero, theirs is rounding noise of another size on every implementation
                nr, floor = float(r.double().norm()), 1e-5 * top
                dn = max(0.0, abs(float(g.double().norm()) - nr) - floor) / max(nr, floor)
                de = max(0.0, float((g - r).abs().max()) - floor) / max(float(r.abs().max()), floor)
                worst_n, worst_e = max(worst_n, dn), max(worst_e, de)
                if dn > rtol or de > 4 * rtol:
                    bad.append("%s rank %d iteration %d: reduced gradient of %s.%s: norm off by %.2e, worst element by %.2e (norm ratio %.4f)"
                               % (mode, rank, it, key, name, dn, de, float(g.norm()) / max(nr, 1e-30)))
        rec["grad_norm_rel_err_" + key], rec["grad_elem_rel_err_" + key] = worst_n, worst_e
    # every parameter after ITERS Adam steps: an element moves by at most lr per step, and its direction is the sign of a gradient
    # that agrees to ~1e-4 - so the UPDATES agree except where the gradient is rounding noise (golden_util.check_checksums' model)
    for key, net, ost in (("d", D, oD), ("g", G, oG)):
        want = dict(zip([n for n, _ in net.named_parameters()], O.trainable(ost)))
        worst_u = 0.0
        for name, p in net.named_parameters():
            got, ref = p.detach().float().cpu(), want[name].detach()
            d = (got - ref).abs()
            if float(d.max()) > 2.02 * ITERS * LR:
                bad.append("%s rank %d: parameter %s.%s differs by %.3e > 2 * steps * lr" % (mode, rank, key, name, float(d.max())))
            dn = abs(float(got.double().norm()) - float(ref.double().norm())) / max(1.0, float(ref.double().norm()))
            worst_u = max(worst_u, dn)
            if dn > 1e-3:
                bad.append("%s rank %d: norm of parameter %s.%s off by %.2e" % (mode, rank, key, name, dn))
        rec["param_norm_rel_err_" + key] = worst_u
    return bad, rec


def rank_main(rank, world, port, out_prefix):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 2) // world)))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    result = {"rank": rank, "failures": []}
    try:
        runs = {}
        for mode, graphed in (("eager", False), ("graph", True)):
            runs[mode] = gpu_run(rank, graphed)
        oracle = oracle_run(rank, world)
        for mode in runs:
            bad, rec = compare(rank, world, mode, runs[mode], oracle)
            result["failures"] += bad
            result[mode] = rec
        # across ranks: identical parameters and spectral-norm vectors (BatchNorm running statistics are rank-local),
        # identical reduced gradients; across launch modes on one rank: everything
        sigs = {}
        for mode, (G, D, reduced, losses) in runs.items():
            shared = [p for net in (G, D) for p in net.parameters()]
            shared += [b for net in (G, D) for n, b in net.named_buffers() if n.endswith("weight_u") or n.endswith("weight_v")]
            sigs[mode] = {"state": digest(shared), "grads": digest([g for key in ("d", "g") for step in reduced[key] for g in step]),
                          "buffers": digest([b for net in (G, D) for b in net.buffers()]), "losses": losses}
        every = [None] * world
        dist.all_gather_object(every, {m: (s["state"], s["grads"]) for m, s in sigs.items()})
        for mode in sigs:
            if len({e[mode] for e in every}) != 1:
                result["failures"].append("%s: ranks disagree on parameters / reduced gradients: %s" % (mode, [e[mode] for e in every]))
        if (sigs["eager"]["state"], sigs["eager"]["grads"], sigs["eager"]["buffers"]) != (sigs["graph"]["state"], sigs["graph"]["grads"], sigs["graph"]["buffers"]):
            result["failures"].append("rank %d: eager and replayed runs differ" % rank)
        if sigs["eager"]["losses"] != sigs["graph"]["losses"]:
            result["failures"].append("rank %d: eager and replayed losses differ" % rank)
        result["ranks_bit_identical"] = all(len({e[m] for e in every}) == 1 for m in sigs)
        import semantic_pyramid_for_image_generation_amd._lib as L
        result["native_library"] = getattr(L.lib(), "_name", None)
    except Exception as exc:                                         # report, never hang the partner in a collective silently
        import traceback
        result["failures"].append("rank %d raised %s: %s\n%s" % (rank, type(exc).__name__, exc, traceback.format_exc()))
    finally:
        with open("%s.rank%d.json" % (out_prefix, rank), "w") as f:
            json.dump(result, f)
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    return 1 if result["failures"] else 0


def launch(out_path, world=2, timeout_s=900.0):
    """Starts the ranks as fresh children (this process never touches the GPU), waits, merges their records into out_path."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    prefix = out_path + ".part"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(r), str(world), str(port), prefix], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    deadline = time.time() + timeout_s
    merged = {"world": world, "ranks": [], "failures": [], "logs": []}
    import signal

    def stop(signum, frame):                                         # the session ends early: take the ranks (exact PIDs) along
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.exit(143)
    signal.signal(signal.SIGTERM, stop)
    while any(p.poll() is None for p in procs) and time.time() < deadline:
        if any(p.poll() not in (None, 0) for p in procs):
            time.sleep(5.0)                                          # a rank failed: give the partner a moment to write its record
            break
        time.sleep(0.5)
    for r, p in enumerate(procs):
        if p.poll() is None:
            merged["failures"].append("rank %d did not finish (killed)" % r)
            p.kill()                                                 # the exact PID started above
        try:
            log = p.communicate(timeout=30)[0]
        except Exception:
            log = ""
        merged["logs"].append(log[-4000:] if log else "")
        try:
            rec = json.load(open("%s.rank%d.json" % (prefix, r)))
            merged["ranks"].append(rec)
            merged["failures"] += rec["failures"]
            os.remove("%s.rank%d.json" % (prefix, r))
        except (OSError, ValueError):
            merged["failures"].append("rank %d left no record (exit code %s)" % (r, p.returncode))
    # bench.py's own N > 1 path (rank spawning, rendezvous, the reducer inside the timed loop, capture agreement, the max-over-ranks
    # clock, comm_stats, the multi_gpu record) has no multi-GPU box to run on either: a dry run with two ranks on this GPU over gloo
    try:
        env = dict(os.environ, BENCH_ONE_DEVICE="1", BENCH_DIST_BACKEND="gloo", BENCH_RANK_TIMEOUT_S="600")
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                              "--channel-factor", "8", "--no-cpu-baseline", "--no-sub-records", "--no-kernel-probe"],
                             env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        merged["bench_two_ranks"] = {"returncode": out.returncode, "line": json.loads(line[-1]) if line else None,
                                     "line_chars": len(line[-1]) if line else 0, "stderr_tail": out.stderr[-1500:]}
    except Exception as exc:
        merged["bench_two_ranks"] = {"returncode": -1, "line": None, "stderr_tail": "%s: %s" % (type(exc).__name__, exc)}
    tmp = out_path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(merged, f, indent=1)
    os.replace(tmp, out_path)
    return 1 if merged["failures"] else 0


if __name__ == "__main__":
    if sys.argv[1] == "launch":
        sys.exit(launch(sys.argv[2]))
    sys.exit(rank_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]))
