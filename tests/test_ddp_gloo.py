"""Multi-process data-parallel contract on CPU (gloo, world_size 2): after the reducer every rank holds the AVERAGE over ranks
of the single-rank gradients computed on its shard with identical parameters.  The arithmetic here is
the CPU oracle - the reducer is what is under test: the flat, in-place path the training step uses (ranges of ONE gradient
buffer handed over group by group, ops.SpectralNormBank.flat) and the per-parameter fallback."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard_grads(rank_shard):
    """Discriminator LSGAN-real gradients of the oracle on one shard (cf=8, batch 2)."""
    from oracle import sempyr_oracle as O
    from semantic_pyramid_for_image_generation_amd import params, synthetic
    D = O.make_state(params.synth_state_dict(O.layout_template(O.discriminator_layout(8)), 1))
    images, labels, _ = synthetic.synthetic_batch(2, 40 + rank_shard, resolution=128)
    pred = O.discriminator_forward(D, images, labels, True)
    O.lsgan_discriminator_loss(pred, pred)[0].backward()
    return D, O.trainable(D)


def _worker(rank, world, port, bucket_bytes, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_pyramid_for_image_generation_amd.distributed import GradientReducer
        D, plist = _shard_grads(rank)
        expected = None
        for r in range(world):                       # every rank recomputes all shards locally as the expectation
            _, pl = _shard_grads(r)
            g = [p.grad.clone() for p in pl]
            expected = g if expected is None else [a + b for a, b in zip(expected, g)]
        expected = [e / world for e in expected]
        GradientReducer(bucket_bytes=bucket_bytes).reduce(plist)
        err = max(float((p.grad - e).abs().max() / (e.abs().max() + 1e-12)) for p, e in zip(plist, expected))
        results[rank] = err
    finally:
        dist.destroy_process_group()


def _flat_worker(rank, world, port, bucket_bytes, results):
    """The training step's path: every gradient of the network is a view of one flat fp32 buffer; ranges of it are reduced in
    place - first the 'groups' that finish early in a backward pass (the tail of the buffer), then whatever is left."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_pyramid_for_image_generation_amd.distributed import GradientReducer
        _, plist = _shard_grads(rank)
        expected = None
        for r in range(world):
            _, pl = _shard_grads(r)
            g = [p.grad.clone() for p in pl]
            expected = g if expected is None else [a + b for a, b in zip(expected, g)]
        expected = [e / world for e in expected]
        offs, total = [], 0
        for p in plist:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total)
        for p, o in zip(plist, offs):
            flat[o:o + p.numel()].copy_(p.grad.flatten())
            p.grad = flat[o:o + p.numel()].view(p.shape)                     # .grad are views: reduced in place
        red = GradientReducer(bucket_bytes=bucket_bytes)
        cut = offs[len(offs) * 3 // 4]
        bucket = max(1, bucket_bytes // 4)
        for a in range(cut, total, bucket):                                  # a late layer group reports first ...
            red.reduce_range(flat, a, min(total, a + bucket))
        red.reduce_flat(flat, [(a, min(cut, a + bucket)) for a in range(0, cut, bucket)])       # ... the rest after backward
        covered = sorted(red.log)
        assert covered[0][0] == 0 and covered[-1][1] == total and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))
        red.join()
        err = max(float((p.grad - e).abs().max() / (e.abs().max() + 1e-12)) for p, e in zip(plist, expected))
        results[rank] = (err, all(p.grad.data_ptr() == flat.data_ptr() + 4 * o for p, o in zip(plist, offs)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 16, 32 << 20])
def test_flat_in_place_reduction_averages_shard_gradients(bucket_bytes):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_flat_worker, args=(world, port, bucket_bytes, results), nprocs=world, join=True)
        assert len(results) == world
        for r in range(world):
            err, in_place = results[r]
            assert err <= 1e-6 and in_place, (r, results[r])


def test_bank_flat_layout_covers_every_parameter_once():
    """ops.SpectralNormBank's flat gradient layout (host logic, no kernels): every parameter of G / D owns exactly one slot,
    the layer groups are contiguous, and flat_ranges() tiles the buffer - the groups that finish first in backward first."""
    import semantic_pyramid_for_image_generation_amd as sp
    for net in (sp.Generator(channels_factor=8), sp.Discriminator(channel_factor=8)):
        bank = net._bank
        bank.set_groups(4)
        bank._build(torch.float32, "cpu")
        bank._alloc_flat("cpu")
        assert len(bank.groups) == 4
        views = bank.w_views + [v for v, (m, _, _) in zip(bank.b_views, bank.specs) if hasattr(m, "bias")] + bank.extra_views
        params = [m.weight_orig for m, _, _ in bank.specs] + [m.bias for m, _, _ in bank.specs if hasattr(m, "bias")] + bank.extra_params
        assert {id(p) for p in params} == {id(p) for p in net.parameters()}
        spans = sorted((v.data_ptr(), v.data_ptr() + 4 * v.numel()) for v in views)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                       # no overlap
        assert spans[0][0] >= bank.flat.data_ptr() and spans[-1][1] <= bank.flat.data_ptr() + 4 * bank.flat_floats
        assert [g[0] for g in bank.group_range[1:]] == [g[1] for g in bank.group_range[:-1]] and bank.group_range[0][0] == 0
        ranges = bank.flat_ranges(1 << 18, min_floats=0)
        assert ranges[0][0] == bank.group_range[-1][0]                                    # the last group leads
        cov = sorted(ranges)
        assert cov[0][0] == 0 and cov[-1][1] == bank.flat_floats and all(x[1] == y[0] for x, y in zip(cov, cov[1:]))
        # no collective below the minimum (default 8 MB) unless the buffer itself is smaller: short spans join their neighbours
        for min_floats in (1 << 16, 1 << 18, 1 << 30):
            ranges = bank.flat_ranges(max(1 << 19, min_floats), min_floats=min_floats)
            cov = sorted(ranges)
            assert cov[0][0] == 0 and cov[-1][1] == bank.flat_floats and all(x[1] == y[0] for x, y in zip(cov, cov[1:]))
            tail = (bank.sn_floats, bank.flat_floats)
            assert len(ranges) <= 2 or all(b - a >= min_floats for a, b in ranges if (a, b) != tail), (min_floats, ranges)
            lead = [r for r in ranges if r[0] == ranges[0][0] or r[0] > ranges[0][0]]
            assert ranges[0][0] > 0 or len(ranges) <= 2, ranges                                           # the late layers still lead
            assert max(b for _, b in lead) in (bank.sn_floats, bank.flat_floats), ranges


def test_bank_rebuild_keeps_the_flat_gradient_buffer():
    """A rebuild of the bank's tables (compute-dtype switch, re-packed weights) must not drop the flat gradient buffer while its
    size is unchanged: captured graphs, the Adam plans and the parameters' .grad views point into it (round-2 ADVICE)."""
    import semantic_pyramid_for_image_generation_amd as sp
    net = sp.Discriminator(channel_factor=8)
    bank = net._bank
    bank._build(torch.float32, "cpu")
    bank._alloc_flat("cpu")
    ptr, floats = bank.flat.data_ptr(), bank.flat_floats
    bank._build(torch.bfloat16, "cpu")
    assert bank.flat is not None and bank.flat.data_ptr() == ptr and bank.flat_floats == floats


def test_bank_regroup_rebuilds_the_gradient_windows():
    """set_groups() changes the per-layer offsets inside a flat buffer of the SAME size (round-3 ADVICE): the retained buffer's
    parameter windows, group counters and stale .grad views must follow the new table, in both directions (4 -> 1, 1 -> 4)."""
    import semantic_pyramid_for_image_generation_amd as sp
    from semantic_pyramid_for_image_generation_amd import _lib as L
    import ctypes
    net = sp.Discriminator(channel_factor=8)
    bank = net._bank

    def table_offsets():
        raw = bytes(bank.bwd_table_dev.numpy().tobytes())
        tab = (L.SpSnBwdLayer * len(bank.specs)).from_buffer_copy(raw)
        return [int(t.grad_off) for t in tab], [int(t.bias_off) for t in tab]
    bank.set_groups(4)
    bank._build(torch.float32, "cpu")
    bank._alloc_flat("cpu")
    ptr = bank.flat.data_ptr()
    probe = bank.specs[3][0].weight_orig
    probe.grad = bank.w_views[3]                          # a gradient left over from the old layout
    for groups in (1, 4, 1):
        bank.set_groups(groups)
        bank._build(torch.float32, "cpu")
        assert bank.flat.data_ptr() == ptr and len(bank.group_count) == groups == len(bank.groups)
        g_off, b_off = table_offsets()
        for i, (m, _, _) in enumerate(bank.specs):
            assert bank.w_views[i].data_ptr() == ptr + 4 * g_off[i], (groups, i)
            assert bank.b_views[i].data_ptr() == ptr + 4 * b_off[i], (groups, i)
        assert probe.grad is None or probe.grad.data_ptr() == bank.w_views[3].data_ptr()
        probe.grad = bank.w_views[3]
        bank.enter_backward(torch.device("cpu"), groups - 1)      # used to raise IndexError after 1 -> 4


def test_bench_spawns_its_own_ranks():
    """bench.py --gpus N without an outer launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and
    rendezvous on 127.0.0.1; the parent (which never touches the GPU) relays rank 0's output and the worst exit code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.argv = ['bench.py', '--gpus', '3']; sys.path.insert(0, %r); import bench; "
            "bench.__file__ = %r; sys.exit(bench.spawn_ranks(3))") % (root, os.path.join(root, "tests", "_rank_echo.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "rank 0 of 3 local 0 master 127.0.0.1", out.stdout


@pytest.mark.parametrize("mode,want", [("die", 7), ("hang", 124)])
def test_bench_rank_supervision(mode, want):
    """A rank that dies takes the job down with ITS exit code; a job that hangs ends at the timeout (code 124) - in both cases
    the remaining ranks are stopped and the call returns (round-2 VERDICT weak #7: `communicate()` without a limit)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.argv = ['bench.py', '--gpus', '2']; sys.path.insert(0, %r); import bench; "
            "bench.__file__ = %r; sys.exit(bench.spawn_ranks(2, timeout_s=3.0))") % (root, os.path.join(root, "tests", "_rank_fail.py"))
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RANK_FAIL_MODE=mode))
    assert out.returncode == want, (out.returncode, out.stderr)
    assert time.time() - t0 < 60 and "stopping the other ranks" in out.stderr


@pytest.mark.parametrize("bucket_bytes", [1 << 16, 32 << 20])
def test_gradient_reducer_averages_shard_gradients(bucket_bytes):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, bucket_bytes, results), nprocs=world, join=True)
        assert len(results) == world
        for r in range(world):
            assert results[r] <= 1e-6, (r, results[r])


def test_reducer_is_identity_without_process_group():
    from semantic_pyramid_for_image_generation_amd.distributed import GradientReducer
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.full((4,), 3.0)
    GradientReducer().reduce([p])
    assert torch.equal(p.grad, torch.full((4,), 3.0))


def _agree_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import semantic_pyramid_for_image_generation_amd as sp
        from semantic_pyramid_for_image_generation_amd.distributed import GradientReducer
        G, D = sp.Generator(channels_factor=16), sp.Discriminator(channel_factor=16)
        mw = sp.ModelWrapper(G, D, None, None, vgg16=sp.VGG16(), save_data_path=None, gradient_reducer=GradientReducer())
        assert mw._reducer_active()
        # one rank's capture failed: every rank must learn it (the eager and the replay path issue different collective sequences)
        results[rank] = (mw._ranks_agree(rank != 1), mw._ranks_agree(True))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_outcome_of_a_graph_capture():
    """Round-3 ADVICE: ModelWrapper.train() decides capture / replay / eager fallback per process; under data parallelism a rank
    that failed to capture must pull every rank back to eager launches (all-reduce(MIN) of the per-rank flag)."""
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_agree_worker, args=(world, port, results), nprocs=world, join=True)
        assert len(results) == world
        for r in range(world):
            assert results[r] == (False, True), (r, results[r])


def _oracle_side_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _two_rank_step as T
        G, D, red_g, red_d, losses = T.oracle_run(rank, world)
        from oracle import sempyr_oracle as O
        results[rank] = (T.digest(O.trainable(G) + O.trainable(D)), T.digest([g for step in red_d + red_g for g in step]),
                         float(red_d[0][0].abs().sum()), losses)
    finally:
        dist.destroy_process_group()


def test_expectation_of_the_two_rank_gpu_job():
    """The oracle side of tests/test_gpu_two_ranks.py (which needs a GPU for its other half): two gloo ranks, each on its own
    shard, with optimizers that average the gradients before Adam - both ranks end with bit-identical parameters and reduced
    gradients although their losses (their shards) differ."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_oracle_side_worker, args=(world, port, results), nprocs=world, join=True)
        assert len(results) == world
        (p0, g0, s0, l0), (p1, g1, s1, l1) = results[0], results[1]
        assert p0 == p1 and g0 == g1 and s0 == s1 and s0 > 0
        assert l0 != l1
