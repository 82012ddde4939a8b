"""N > 1 path.

CPU tier : world_size-2 gloo test of the host-side sharding logic (shard partition of
           the seeded stream, all-reduce window arithmetic of the plan's communication
           buffer) -- the parts of the multi-GPU fit that do not need a GPU.
GPU tier : the full sharded fit with 2 ranks sharing the one GPU of the test box (gloo
           backend on device tensors): coefficients match the reference golden vector
           and are bit-identical on both ranks.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT, load_golden, relmax


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from splpak_amd.dist import make_allreduce, shard_range
    from splpak_amd.synth import synth_points
    _init(rank, world, port)
    try:
        # 1. shards partition the stream: concatenating the ranks' slices gives the global data
        nd, m = 3, 1001
        first, cnt = shard_range(m, rank, world)
        x, y, w = synth_points(nd, cnt, first_point=first)
        xg, yg, wg = synth_points(nd, m)
        ok = np.array_equal(x, xg[first:first + cnt]) and np.array_equal(y, yg[first:first + cnt]) \
            and np.array_equal(w, wg[first:first + cnt])
        # 2. window arithmetic of the communication buffer (layout of plan.hip: G | H | R)
        ncol, h, npad = 512, 172, 512
        lenG, lenH, lenR = ncol * h + ncol + 8, ncol + 8, npad
        comm = torch.zeros(lenG + lenH + lenR, dtype=torch.float64)
        ar = make_allreduce(comm, dist)
        comm[lenG:lenG + lenH] = float(rank + 1)               # histogram region
        ar(lenG, lenH)
        ok = ok and bool((comm[lenG:lenG + lenH] == 3.0).all()) and float(comm[:lenG].abs().sum()) == 0.0
        comm[:lenG] = torch.arange(lenG, dtype=torch.float64) * (rank + 1)
        ar(0, lenG)
        ok = ok and bool((comm[:lenG] == torch.arange(lenG, dtype=torch.float64) * 3).all())
        ok = ok and float(comm[lenG + lenH:].abs().sum()) == 0.0
        try:
            ar(lenG + lenH, lenR + 1)
            ok = False
        except ValueError:
            pass
        q.put((rank, ok, cnt))
    finally:
        dist.destroy_process_group()


def test_sharding_logic_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, 29611, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert sum(c for _, _, c in res) == 1001


def test_shard_range_covers_everything():
    from splpak_amd.dist import shard_range
    for n, w in [(10, 3), (7, 8), (10**8, 8), (1, 1)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1


def _gpu_worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    from splpak_amd.dist import ShardedFit, shard_range
    from tests.cases import CASES, make_inputs
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        inp = make_inputs(CASES[name])
        m = inp["xdata"].shape[0]
        first, cnt = shard_range(m, rank, world)
        x = torch.tensor(inp["xdata"][first:first + cnt], device=dev)
        y = torch.tensor(inp["ydata"][first:first + cnt], device=dev)
        w = None if inp["wdata"] is None else torch.tensor(inp["wdata"][first:first + cnt], device=dev)
        ncol = int(np.prod(inp["nodes"]))
        coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
        sf = ShardedFit(inp["ndim"], inp["nodes"], inp["xmin"], inp["xmax"], inp["xtrap"], max(cnt, 1), dev, dist)
        ierr, info = sf.fit(x, y, w, coef, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        first_coef = coef.clone()
        # the same plan again, twice: identical bits (round 3: the reductions of later fits summed buffers that the
        # fit's kernels were still writing -- seen in `bench.py --gpus 2` on one device, never in a single fit)
        for _ in range(2):
            ierr2, _ = sf.fit(x, y, w, coef, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert ierr2 == ierr and torch.equal(coef, first_coef), "repeated sharded fits differ"
        q.put((rank, ierr, coef.cpu().numpy(), info))
        sf.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["3d8", "2d16_zero_w", "2d8_cc", "3d12"])
def test_sharded_fit_two_ranks_one_gpu(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, 29621, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    gold = load_golden(name)
    for rank, ierr, coef, info in res:
        assert ierr == 0
        assert relmax(coef, gold["coef"]) < 1e-10
    # every rank factors bit-identical normal equations -> identical coefficients
    assert np.array_equal(res[0][2], res[1][2])
    # rows were counted globally
    spec_rows = res[0][3][0]
    assert spec_rows == res[1][3][0]


def _gpu_worker_pcg(rank, world, port, q):
    """4-D 12^4 at config 5's density of points, the ITERATIVE solve alone (round 6), points sharded over the ranks."""
    sys.path.insert(0, ROOT)
    os.environ["SPLPAK_SOLVER"] = "pcg"
    from splpak_amd import capi
    from splpak_amd.dist import ShardedFit, shard_range
    from splpak_amd.synth import synth_points
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        nd, nod, m = 4, 12, 158122
        xh, yh, wh = synth_points(nd, m)
        first, cnt = shard_range(m, rank, world)
        x = torch.tensor(xh[first:first + cnt], device=dev)
        y = torch.tensor(yh[first:first + cnt], device=dev)
        w = torch.tensor(wh[first:first + cnt], device=dev)
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        sf = ShardedFit(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, cnt, dev, dist if world > 1 else None)
        ierr, info = sf.fit(x, y, w, coef, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        q.put((rank, ierr, coef.cpu().numpy(), info, sf.plan.factorisation()[0], sf.plan.pcg_stats()))
        sf.close()
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_iterative_solve_two_ranks_one_gpu():
    """The iterative solve with the points sharded over two processes (one GPU here; route (a) of DESIGN section 5): every rank applies
    the rows of ITS points, one all-reduce of the product per iteration, and the preconditioner's two moments travel with the
    histogram's reduction so that every rank builds the same one -- the ranks end with identical coefficients, 1e-10 from the
    single-process fit of all points (the sums over the points are ordered differently) and from the factorisation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_pcg, args=(r, 2, 29671, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, ierr, coef, info, code, ps in res:
        assert ierr == 0 and code == 6 and ps["iterations"] > 0 and info[9] < 1e-9, (rank, ierr, code, ps)
    assert np.array_equal(res[0][2], res[1][2])
    assert res[0][5]["iterations"] == res[1][5]["iterations"]
    from splpak_amd import capi
    from splpak_amd.synth import synth_points
    xh, yh, wh = synth_points(4, 158122)
    os.environ["SPLPAK_SOLVER"] = "direct"
    try:
        c1, e1, _, _ = capi.fit(4, xh, yh, wh, [0.0] * 4, [1.0] * 4, [12] * 4, 1.0)
    finally:
        os.environ.pop("SPLPAK_SOLVER", None)
    assert e1 == 0
    print(f"two ranks, {res[0][5]['iterations']} iterations: {relmax(res[0][2], c1):.2e} from the single-GPU factorisation")
    assert relmax(res[0][2], c1) < 1e-10


def _gpu_worker_nd(rank, world, port, name, q):
    os.environ["SPLPAK_ND"] = "1"            # (the goldens' grids are below the size from which nested dissection is the default)
    os.environ.pop("SPLPAK_ND_DIST", None)
    _gpu_worker(rank, world, port, name, q)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("3d12", 2), ("3d16", 2), ("3d16", 3), ("3d16", 4), ("2d64_c2grid", 4)])
def test_sharded_fit_with_the_factorisation_distributed_by_subtrees(name, world):
    """Round 3 (SPLPAK_ND_DIST=0 turns it off): the ranks of a sharded fit eliminate their own subtrees of the nested-dissection tree only;
    the Schur complements they leave in the fronts of depth dcut - 1 are summed through the all-reduce hook, the top of the
    tree is factored by every rank, and the tree solves exchange the same fronts' vectors and the solution.  Rehearsed
    with 2 - 4 ranks on ONE GPU over gloo (there is no multi-GPU hardware in this pool): the reference's golden
    coefficients at 1e-10, identical bits on every rank, repeated fits identical (in _gpu_worker)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_nd, args=(r, world, 29631 + world, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    gold = load_golden(name)
    for rank, ierr, coef, info in res:
        assert ierr == 0
        assert relmax(coef, gold["coef"]) < 1e-10
        assert np.array_equal(coef, res[0][2])


def _gpu_worker_unbalanced(rank, world, port, nd, nodes, two_arg_hook, q):
    """Sharded fit on a grid whose nested-dissection tree is UNBALANCED (a split straddles split_min: some subtrees stop one
    level short), three fits through one plan, compared with the unsharded fit of the same points."""
    os.environ["SPLPAK_ND"] = "1"
    os.environ.pop("SPLPAK_ND_DIST", None)
    sys.path.insert(0, ROOT)
    from splpak_amd import capi
    from splpak_amd.dist import ShardedFit, make_allreduce, shard_range
    from splpak_amd.synth import synth_points
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ncol = int(np.prod(nodes))
        m = 6 * ncol
        x, y, w = synth_points(nd, m)
        lo, hi = [0.0] * nd, [1.0] * nd
        first, cnt = shard_range(m, rank, world)
        xs, ys, ws = (torch.tensor(a[first:first + cnt], device=dev) for a in (x, y, w))
        coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
        sf = ShardedFit(nd, nodes, lo, hi, 1.0, max(cnt, 1), dev, dist)
        if two_arg_hook:            # a hook written to the contract of rounds 1-2: windows of the plan's buffer only
            full = make_allreduce(sf.comm, dist)
            sf.plan.set_allreduce(lambda off, cnt_: full(off, cnt_), rank, world)
        st = torch.cuda.current_stream().cuda_stream
        outs = []
        for _ in range(3):
            ierr, info = sf.fit(xs, ys, ws, coef, st)
            torch.cuda.synchronize()
            outs.append((ierr, coef.cpu().numpy().copy()))
        ref = None
        if rank == 0:
            ref, e1, _, _ = capi.fit(nd, x, y, w, lo, hi, nodes, 1.0)
            assert e1 == 0
        q.put((rank, outs, ref, info))
        sf.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("nd,nodes,world,two_arg", [(3, [18, 18, 18], 2, False), (2, [34, 34], 2, False), (3, [18, 18, 18], 3, False),
                                                    (3, [12, 12, 12], 2, True)])
def test_distributed_subtrees_unbalanced_tree_repeated_fits(nd, nodes, world, two_arg):
    """Round-3 advice (high): on an unbalanced tree a rank whose subtrees stop short of the tree's depth skipped the stage
    bookkeeping of the depths it has no front at, and its second and later fits added onto the previous fit's Schur
    buffers.  18^3 with 2 ranks: rank 0 holds depths 1..5 of a depth-6 tree.  Every fit of the plan must give the same
    bits, on every rank, and agree with the unsharded fit.  The two-argument hook of rounds 1-2 (no SPLPAK_AR_ANY_POINTER)
    must keep working through the replicated factorisation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_unbalanced, args=(r, world, 29651 + world, nd, nodes, two_arg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    ref = res[0][2]
    for rank, outs, _, info in res:
        for ierr, c in outs:
            assert ierr == 0
            assert np.array_equal(c, res[0][1][0][1]), "fits differ between ranks or between repeated fits"
        assert relmax(outs[0][1], ref) < 1e-11
        assert info[9] < 1e-9


def _gpu_worker_nd_singular(rank, world, port, q):
    os.environ["SPLPAK_ND"] = "1"
    os.environ.pop("SPLPAK_ND_DIST", None)
    sys.path.insert(0, ROOT)
    from splpak_amd.dist import ShardedFit, shard_range
    from splpak_amd.synth import synth_points
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        nd, nod = 3, 12
        x, y, w = synth_points(nd, 3000)
        x = x * 0.5                                  # an empty half of the box: columns without data, no smoothing rows
        first, cnt = shard_range(x.shape[0], rank, world)
        xs, ys, ws = (torch.tensor(a[first:first + cnt], device=dev) for a in (x, y, w))
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        sf = ShardedFit(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 0.0, max(cnt, 1), dev, dist)
        ierr, info = sf.fit(xs, ys, ws, coef, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        q.put((rank, ierr))
        sf.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_distributed_subtrees_singular_system_is_107_on_every_rank():
    """A pivot fails in ONE rank's subtree: every rank must come back with 107 (the failure reaches the top of the tree through
    the summed Schur complements, and the pivot status is all-reduced) -- none may be left waiting in a collective."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_nd_singular, args=(r, world, 29641, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
    assert [e for _, e in res] == [107] * world


# ---------------------------------------------------------------------------
# distributed band (one process, several GPUs): rehearsed with virtual GPUs on the one device
# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,ngpus,chunk", [("3d12", 2, 1), ("3d12", 3, 2), ("2d64_c2grid", 4, 1), ("4d6", 2, 1),
                                              ("3d8_cc_clust", 4, 1), ("2d16_zero_w", 3, 1), ("c1_1d16", 2, 1),
                                              ("2d64_c2grid", 1, 1), ("3d12", 1, 2)])     # one rank: the distributed code on a plan that would otherwise be two-ended
def test_distributed_band_fit_virtual_gpus(name, ngpus, chunk, monkeypatch):
    """splpak_fit_multi_f64 with every rank on the one GPU of the test box (SPLPAK_VIRTUAL_GPUS): point
    shards, rank-ordered reductions, block columns dealt to the ranks, panel hand-over, look-ahead and
    the owner-to-owner sweeps all run; the result must hold the reference golden at 1e-10 and agree
    with the single-GPU fit."""
    from splpak_amd import capi
    from tests.cases import CASES, make_inputs
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    monkeypatch.setenv("SPLPAK_DIST_CHUNK", str(chunk))
    inp = make_inputs(CASES[name])
    gold = load_golden(name)
    args = (inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"], inp["xtrap"])
    c1, e1, h1, i1 = capi.fit(*args, want_hist=True)
    cm, em, hm, im = capi.fit_multi(ngpus, *args, want_hist=True)
    assert e1 == em == 0
    print(f"{name} x{ngpus} (chunk {chunk}): rel={relmax(cm, gold['coef']):.2e} vs single={relmax(cm, c1):.2e} "
          f"steps={im[2]:.0f} optimality={im[9]:.1e}")
    assert relmax(cm, gold["coef"]) < 1e-10
    assert relmax(cm, c1) < 1e-12
    assert im[0] == i1[0] and im[1] == i1[1]                 # rows counted over all shards
    assert im[9] < 1e-9
    if inp["xtrap"] != 0.0:
        assert relmax(hm, gold["hist"]) < 1e-12


@pytest.mark.gpu
def test_distributed_band_24cubed_and_errors(monkeypatch):
    """24^3 grid (54 block steps, band 8 blocks wide) on 4 virtual GPUs: spline-space data reproduced to
    1e-10; a rank-deficient problem returns 107 on every rank (no rank is left waiting)."""
    from splpak_amd import capi
    from splpak_amd.synth import synth_points, synth_queries
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    nd, nod, m = 3, 24, 200000
    x, _, _ = synth_points(nd, m)
    f = lambda p: 1.0 + 2.0 * p[:, 0] - 3.0 * p[:, 1] + 0.5 * p[:, 2]
    c, ierr, _, info = capi.fit_multi(4, nd, x, f(x), None, [0.0] * nd, [1.0] * nd, [nod] * nd, 0.0)
    assert ierr == 0
    q = synth_queries(nd, 2000, m) * 1.2 - 0.1
    v, _ = capi.evaluate(nd, q, None, c, [0.0] * nd, [1.0] * nd, [nod] * nd)
    assert np.max(np.abs(v - f(q))) < 1e-10 and info[9] < 1e-9
    xc = np.random.default_rng(0).random((400, 2)) * 0.2     # all data in one corner, no smoothing rows
    assert capi.fit_multi(3, 2, xc, xc.sum(axis=1), None, [0.0, 0.0], [1.0, 1.0], [8, 8], 0.0)[1] == 107
    assert capi.fit_multi(2, 0, xc, xc.sum(axis=1), None, [0.0, 0.0], [1.0, 1.0], [8, 8], 0.0)[1] == 101


@pytest.mark.gpu
def test_distributed_band_4d_12_property(monkeypatch):
    """BASELINE config 5's shape at a size one test can afford: 4-D 12^4 grid (20 736 columns, band 23
    blocks wide, 81 block steps) through the distributed path on 4 virtual GPUs.  Data sampled from a
    spline of the grid with random coefficients (xtrap = 0: the fit is a projection) must give those
    coefficients back to 1e-10 -- the dense reference cannot run this grid in test time."""
    import torch
    from splpak_amd import capi
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    nd, nod, m, R = 4, 12, 400000, 4
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    ctrue = torch.randn(nod ** nd, dtype=torch.float64, device=dev, generator=gen)
    per = m // R
    xs, ys = [], []
    for r in range(R):
        x = torch.empty((per, nd), dtype=torch.float64, device=dev)
        capi.synth_points_dev(nd, r * per, per, x, None, None, st)
        y = torch.empty(per, dtype=torch.float64, device=dev)
        capi.evaluate_dev(nd, x, None, ctrue, lo, hi, nodes, y, st)
        xs.append(x)
        ys.append(y)
    torch.cuda.synchronize()
    mp = capi.MultiPlan(R, nd, nodes, lo, hi, 0.0, per, chunk=2)
    try:
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        ierr, info = mp.fit(xs, ys, None, coef)
        err = float((coef - ctrue).abs().max() / ctrue.abs().max())
        print(f"4-D 12^4 x{R} virtual GPUs: coefficient error {err:.2e}, steps {info[2]:.0f}, optimality {info[9]:.1e}, "
              f"phases {info[5]:.3f} / {info[6]:.3f} / {info[7]:.3f} s")
        assert ierr == 0 and info[0] == m
        assert err < 1e-10 and info[9] < 1e-9
    finally:
        mp.close()


# ---------------------------------------------------------------------------
# nested dissection distributed over the GPUs of one process (round 4): subtrees per GPU, top fronts by block columns
# ---------------------------------------------------------------------------
def _mplan_fit(inp, ngpus, shard, coef_dev=None):
    """Fit through splpak_mplan_* with every rank on the one GPU of the box.  shard=False: rank 0 holds every point (the
    normal equations are then bit-identical to the single-GPU fit's)."""
    from splpak_amd import capi
    dev = torch.device("cuda", 0)
    m = inp["xdata"].shape[0]
    nd = inp["ndim"]
    bounds = [(r * m) // ngpus for r in range(ngpus + 1)] if shard else [0] + [m] * ngpus
    xs = [torch.tensor(inp["xdata"][bounds[r]:bounds[r + 1]].reshape(-1, nd), device=dev) for r in range(ngpus)]
    ys = [torch.tensor(inp["ydata"][bounds[r]:bounds[r + 1]], device=dev) for r in range(ngpus)]
    ws = None if inp["wdata"] is None else [torch.tensor(inp["wdata"][bounds[r]:bounds[r + 1]], device=dev) for r in range(ngpus)]
    ncol = int(np.prod(inp["nodes"]))
    coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
    mpl = capi.MultiPlan(ngpus, nd, inp["nodes"], inp["xmin"], inp["xmax"], inp["xtrap"], max(m, 1))
    try:
        outs = []
        for _ in range(2):                      # twice through one plan: identical bits
            ierr, info = mpl.fit(xs, ys, ws, coef)
            outs.append(coef.cpu().numpy().copy())
        assert np.array_equal(outs[0], outs[1]), "repeated multi-GPU fits differ"
        return outs[0], ierr, info
    finally:
        mpl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,ngpus,chunk", [("3d12", 2, 1), ("3d16", 2, 1), ("3d16", 3, 1), ("3d16", 4, 2), ("3d16", 8, 1),
                                              ("2d64_c2grid", 4, 1), ("2d64_c2grid", 8, 1), ("3d8_cc_clust", 2, 1),
                                              ("2d16", 8, 1), ("2d16_sparse", 3, 1), ("3d_aniso", 6, 1),     # more ranks than subtrees (a tree of 4 leaves on 8 ranks), odd rank counts
                                              # 4-D goldens (the configuration this route exists for is 4-D).  Their trees
                                              # have ONE front at the default leaf size (4^4 at any), so the leaves are made as small
                                              # as the tree builder allows (SPLPAK_ND_SPLIT=5: 5^4 and 6^4 -> 31 fronts of
                                              # 1 .. 3 nodes per dimension) -- for the single-GPU fit they are compared with as well
                                              ("4d5_cc", 2, 1), ("4d5_cc", 4, 1), ("4d6", 2, 1), ("4d6", 4, 2), ("4d6", 8, 1)])
def test_multi_gpu_nested_dissection_is_bitwise_the_single_gpu_fit(name, ngpus, chunk, monkeypatch):
    """splpak_mplan_* (what Fortran's set_gpus reaches) factor through the nested-dissection tree -- every rank
    stores and eliminates only its subtrees, the fronts above are distributed by block columns with peer-copied panels and
    pulled Schur complements.  With every point on rank 0 the normal equations are those of the single-GPU fit, and so must
    be every bit of the coefficients (same operations per element in the same order); with the points sharded the sums of
    the normal equations are ordered differently: 1e-12.  Goldens at 1e-10 either way."""
    from splpak_amd import capi
    from tests.cases import CASES, make_inputs
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    monkeypatch.setenv("SPLPAK_ND", "1")
    monkeypatch.setenv("SPLPAK_ND_CHUNK", str(chunk))
    if name.startswith("4d"):
        monkeypatch.setenv("SPLPAK_ND_SPLIT", "5")
        assert capi.debug_nd_tree(CASES[name]["nodes"], split_min=5)["fronts"] >= 31
    inp = make_inputs(CASES[name])
    gold = load_golden(name)
    args = (inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"], inp["xtrap"])
    c1, e1, h1, i1 = capi.fit(*args, want_hist=True)
    assert e1 == 0
    c0, e0, info0 = _mplan_fit(inp, ngpus, shard=False)
    cs, es, infos = _mplan_fit(inp, ngpus, shard=True)
    print(f"{name} x{ngpus} (chunk {chunk}): golden rel {relmax(c0, gold['coef']):.2e}; all points on rank 0 vs single GPU: "
          f"{'identical bits' if np.array_equal(c0, c1) else 'rel %.2e' % relmax(c0, c1)}; sharded vs single {relmax(cs, c1):.2e}")
    assert e0 == es == 0
    assert relmax(c0, gold["coef"]) < 1e-10 and relmax(cs, gold["coef"]) < 1e-10
    assert np.array_equal(c0, c1)
    assert relmax(cs, c1) < 1e-12
    assert infos[0] == i1[0] and infos[1] == i1[1] and infos[9] < 1e-9


@pytest.mark.gpu
def test_multi_gpu_nested_dissection_fortran_entry_and_errors(monkeypatch):
    """splpak_fit_multi_f64 (the entry Fortran's set_gpus binds) through the distributed nested dissection: golden + histogram;
    a rank-deficient problem is 107 on every rank; the grid checks come first."""
    from splpak_amd import capi
    from tests.cases import CASES, make_inputs
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    monkeypatch.setenv("SPLPAK_ND", "1")
    inp = make_inputs(CASES["3d12"])
    gold = load_golden("3d12")
    args = (inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"], inp["xtrap"])
    cm, em, hm, im = capi.fit_multi(4, *args, want_hist=True)
    assert em == 0 and relmax(cm, gold["coef"]) < 1e-10 and relmax(hm, gold["hist"]) < 1e-12 and im[9] < 1e-9
    from splpak_amd.synth import synth_points
    x, y, w = synth_points(3, 3000)
    assert capi.fit_multi(3, 3, x * 0.5, y, w, [0.0] * 3, [1.0] * 3, [12] * 3, 0.0)[1] == 107        # an empty half of the box
    assert capi.fit_multi(2, 0, x, y, w, [0.0] * 3, [1.0] * 3, [12] * 3, 0.0)[1] == 101


@pytest.mark.gpu
def test_multi_gpu_nested_dissection_4d_12_property(monkeypatch):
    """BASELINE config 5's shape at a size one test can afford: 4-D 12^4 (20 736 columns) on 4 virtual GPUs through the
    distributed nested dissection -- data sampled from a spline of the grid with random coefficients gives them back."""
    from splpak_amd import capi
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    monkeypatch.setenv("SPLPAK_ND", "1")
    nd, nod, m, R = 4, 12, 400000, 4
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    ctrue = torch.randn(nod ** nd, dtype=torch.float64, device=dev, generator=gen)
    per = m // R
    xs, ys = [], []
    for r in range(R):
        x = torch.empty((per, nd), dtype=torch.float64, device=dev)
        capi.synth_points_dev(nd, r * per, per, x, None, None, st)
        y = torch.empty(per, dtype=torch.float64, device=dev)
        capi.evaluate_dev(nd, x, None, ctrue, lo, hi, nodes, y, st)
        xs.append(x)
        ys.append(y)
    torch.cuda.synchronize()
    mpl = capi.MultiPlan(R, nd, nodes, lo, hi, 0.0, per)
    try:
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        ierr, info = mpl.fit(xs, ys, None, coef)
        err = float((coef - ctrue).abs().max() / ctrue.abs().max())
        print(f"4-D 12^4 x{R} virtual GPUs, nested dissection: coefficient error {err:.2e}, steps {info[2]:.0f}, optimality {info[9]:.1e}, "
              f"phases {info[5]:.3f} / {info[6]:.3f} / {info[7]:.3f} s")
        assert ierr == 0 and info[0] == m
        assert err < 1e-10 and info[9] < 1e-9
    finally:
        mpl.close()


def _single_and_multi(nd, nodes, m, R, weighted=True, xtrap=1.0, shard=False):
    """The same points through a single-GPU plan and through splpak_mplan_* on R virtual GPUs (all points on rank 0 unless
    shard).  -> (single coefficients, multi coefficients, multi info, MultiPlan factorisation code, per-rank bytes, single plan bytes)"""
    from splpak_amd import capi
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    lo, hi = [0.0] * nd, [1.0] * nd
    ncol = int(np.prod(nodes))
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    torch.cuda.synchronize()
    c1 = torch.zeros(ncol, dtype=torch.float64, device=dev)
    plan = capi.Plan(nd, nodes, lo, hi, xtrap, m)
    try:
        e1, i1 = plan.fit(x, y, w if weighted else None, c1, st)
        torch.cuda.synchronize()
        single_bytes = plan.device_bytes()
    finally:
        plan.close()
    assert e1 == 0
    bounds = [(r * m) // R for r in range(R + 1)] if shard else [0] + [m] * R
    xs = [x[bounds[r]:bounds[r + 1]] for r in range(R)]
    ys = [y[bounds[r]:bounds[r + 1]] for r in range(R)]
    ws = [w[bounds[r]:bounds[r + 1]] for r in range(R)] if weighted else None
    cm = torch.zeros(ncol, dtype=torch.float64, device=dev)
    mpl = capi.MultiPlan(R, nd, nodes, lo, hi, xtrap, max(bounds[r + 1] - bounds[r] for r in range(R)))
    try:
        code, what = mpl.factorisation()
        rb = [mpl.rank_bytes(r) for r in range(R)]
        em, im = mpl.fit(xs, ys, ws, cm)
        first = cm.clone()
        em2, _ = mpl.fit(xs, ys, ws, cm)
        assert em == em2 == 0 and torch.equal(first, cm), "repeated multi-GPU fits differ"
    finally:
        mpl.close()
    return c1.cpu().numpy(), cm.cpu().numpy(), im, i1, code, rb, single_bytes


@pytest.mark.gpu
@pytest.mark.parametrize("nd,nodes,m,R", [(3, [32] * 3, 300000, 4), (3, [24, 40, 24], 200000, 3), (2, [150, 130], 200000, 8),
                                          (3, [18] * 3, 60000, 2), (3, [18] * 3, 60000, 3), (2, [66, 70], 50000, 8), (3, [16, 16, 17], 40000, 5)])
def test_multi_gpu_nested_dissection_medium_grids_bitwise(nd, nodes, m, R, monkeypatch):
    """Grids whose top fronts take many block steps (32^3: root of 12 steps, borders of several blocks; an anisotropic box on
    THREE ranks; a 2-D grid on eight; UNBALANCED trees -- 18^3, 66 x 70 -- whose subtrees stop at different depths; five ranks):
    nested dissection is the default from 4 096 columns on, also for the multi-GPU plan."""
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    c1, cm, im, i1, code, rb, sb = _single_and_multi(nd, nodes, m, R)
    print(f"{nodes} x{R}: factorisation code {code}; per-rank bytes {[round(b / 1e9, 2) for b in rb]} GB (single-GPU plan {sb / 1e9:.2f} GB); "
          f"{'identical bits' if np.array_equal(c1, cm) else 'rel %.2e' % relmax(cm, c1)}")
    assert code == 5
    assert np.array_equal(c1, cm)
    assert im[0] == i1[0] and im[1] == i1[1] and im[9] < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 4])
def test_multi_gpu_nested_dissection_c3_full_size(R, monkeypatch):
    """BASELINE config 3 at full size (64^3 nodes, 1e7 weighted points of the seeded stream, xtrap = 1) through the multi-GPU
    plan on R virtual GPUs: every bit of the single-GPU fit's coefficients, and every rank holds only its share of the factor
    (the byte counts the host-only partition promises)."""
    from splpak_amd import capi
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    nodes = [64] * 3
    c1, cm, im, i1, code, rb, sb = _single_and_multi(3, nodes, 10_000_000, R)
    ranks, summ = capi.debug_nd_partition(nodes, R)
    print(f"C3 x{R} virtual GPUs: {'identical bits' if np.array_equal(c1, cm) else 'rel %.2e' % relmax(cm, c1)}; per-rank bytes "
          f"{[round(b / 1e9, 1) for b in rb]} GB, of which factorisation {[round(r['bytes'] / 1e9, 1) for r in ranks]} GB; single-GPU plan "
          f"{sb / 1e9:.1f} GB; phases {im[5]:.3f} / {im[6]:.3f} / {im[7]:.3f} s; backward error {im[9]:.1e}")
    assert code == 5 and np.array_equal(c1, cm)
    assert im[9] < 1e-9
    for r in range(R):
        # what the rank holds = its share of the factorisation + normal equations, binning scratch for its points, Gram scratch, staging
        assert rb[r] > ranks[r]["bytes"] * 0.98
        assert rb[r] < ranks[r]["bytes"] + 2.2 * summ["normal_eq_bytes"] + 4.5e9 + 10_000_000 * 60
    assert max(rb) < 0.75 * sb


@pytest.mark.gpu
def test_multi_gpu_nested_dissection_4d_24_on_four_virtual_gpus(monkeypatch):
    """4-D 24^4 nodes (331 776 columns; 208 GB of panels + Schur arenas on one GPU) on FOUR virtual GPUs,
    memory PARTITIONED: every rank's device bytes stay below a cap derived from the host-only partition (its subtrees + its
    block columns of the top fronts + the replicated normal equations), far below the single-GPU plan's.  Data sampled from
    a spline of the grid with random coefficients (xtrap = 0) gives the coefficients back."""
    from splpak_amd import capi
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    capi.shutdown()
    torch.cuda.empty_cache()
    nd, nod, m, R = 4, 24, 2_000_000, 4
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    ctrue = torch.randn(nod ** nd, dtype=torch.float64, device=dev, generator=gen)
    per = m // R
    xs, ys = [], []
    for r in range(R):
        x = torch.empty((per, nd), dtype=torch.float64, device=dev)
        capi.synth_points_dev(nd, r * per, per, x, None, None, st)
        y = torch.empty(per, dtype=torch.float64, device=dev)
        capi.evaluate_dev(nd, x, None, ctrue, lo, hi, nodes, y, st)
        xs.append(x)
        ys.append(y)
    torch.cuda.synchronize()
    ranks, summ = capi.debug_nd_partition(nodes, R)
    tree = capi.debug_nd_tree(nodes, check=False)
    mpl = capi.MultiPlan(R, nd, nodes, lo, hi, 0.0, per)
    try:
        code, _ = mpl.factorisation()
        rb = [mpl.rank_bytes(r) for r in range(R)]
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        ierr, info = mpl.fit(xs, ys, None, coef)
        err = float((coef - ctrue).abs().max() / ctrue.abs().max())
        print(f"4-D 24^4 x{R} virtual GPUs: per-rank bytes {[round(b / 1e9, 1) for b in rb]} GB (factorisation "
              f"{[round(r['bytes'] / 1e9, 1) for r in ranks]} GB; one GPU would hold {(tree['factor_bytes'] + tree['arena_bytes']) / 1e9:.0f} GB); "
              f"coefficient error {err:.2e}, steps {info[2]:.0f}, optimality {info[9]:.1e}, phases {info[5]:.2f} / {info[6]:.2f} / {info[7]:.2f} s")
        assert code == 5 and ierr == 0 and info[0] == m
        assert err < 1e-10 and info[9] < 1e-9
        for r in range(R):
            cap = ranks[r]["bytes"] + 2.2 * summ["normal_eq_bytes"] + 3e9
            assert rb[r] < cap, f"rank {r} holds {rb[r] / 1e9:.1f} GB, cap {cap / 1e9:.1f} GB"
        assert max(rb) < 0.4 * (tree["factor_bytes"] + tree["arena_bytes"])
    finally:
        mpl.close()
        capi.shutdown()
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_c4_full_size_rehearsed_on_eight_virtual_gpus(monkeypatch):
    """BASELINE config 4 at full size -- 1e8 weighted points of the seeded stream on the 64^3 grid, sharded
    over EIGHT ranks (1.25e7 each) of the one-process multi-GPU plan, nested dissection distributed (subtree per rank, top
    fronts by block columns) -- rehearsed on the one GPU of the test box.  Held to the single-GPU fit of the same 1e8 points."""
    from splpak_amd import capi
    monkeypatch.setenv("SPLPAK_VIRTUAL_GPUS", "1")
    capi.shutdown()
    torch.cuda.empty_cache()
    c1, cm, im, i1, code, rb, sb = _single_and_multi(3, [64] * 3, 100_000_000, 8, shard=True)
    print(f"C4 (1e8 points, 64^3) on 8 virtual GPUs: rel to the single-GPU fit {relmax(cm, c1):.2e}; per-rank bytes "
          f"{[round(b / 1e9, 1) for b in rb]} GB (single-GPU plan {sb / 1e9:.1f} GB); rows {im[0]:.0f}+{im[1]:.0f}; backward error {im[9]:.1e}; "
          f"phases {im[5]:.3f} / {im[6]:.3f} / {im[7]:.3f} s")
    assert code == 5
    assert relmax(cm, c1) < 1e-10
    assert im[0] == i1[0] == 100_000_000 and im[1] == i1[1]
    assert im[9] < 1e-9 and abs(im[8] - i1[8]) <= 1e-9 * i1[8]
    assert max(rb) < 0.5 * sb


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["3d12", "2d64_c2grid"])
def test_stream_ordered_hook_with_two_ranks_is_bitwise_the_synchronised_one(name, monkeypatch):
    """SPLPAK_AR_STREAM_ORDERED (what the native RCCL hook declares: the library does NOT synchronise the fit's
    stream around the hook) had only ever run with a one-rank communicator.  Two ranks here -- two plans on one GPU, a host
    thread each, as two processes would be -- with a hook that only ENQUEUES on the stream it is handed: its buffer goes to a
    staging slot (async copy), an event says so, the peer's event is awaited ON THE STREAM, the sum rank 0 + rank 1 is formed
    by a kernel.  The host threads only meet at a barrier (so that the event a stream waits for has been recorded), never wait
    for the device.  Points sharded, nested dissection distributed by subtrees (any-pointer hook): the coefficients must be
    bit-identical on both ranks, to the same hook run with the library's synchronisations around it, and hold the golden."""
    import threading
    import torch
    from splpak_amd import capi
    from tests.cases import CASES, make_inputs
    monkeypatch.setenv("SPLPAK_ND", "1")
    inp = make_inputs(CASES[name])
    gold = load_golden(name)
    nd, m = inp["ndim"], inp["xdata"].shape[0]
    dev = torch.device("cuda", 0)
    ncol = int(np.prod(inp["nodes"]))

    def run(stream_ordered):
        bar = threading.Barrier(2)
        stage = {}                                   # (generation parity, rank) -> staging tensor
        events = {}
        results, errors = [None, None], []

        def make_hook(rank, comm):
            gen = [0]

            def hook(off, count, view=None):
                v = comm[off:off + count] if view is None else view
                g = gen[0]
                gen[0] += 1
                key = (g & 1, rank)
                if key not in stage or stage[key].numel() < count:
                    stage[key] = torch.empty(max(count, 1), dtype=torch.float64, device=dev)
                stage[key][:count].copy_(v, non_blocking=True)        # on the library's stream (made current by the shim)
                ev = torch.cuda.Event()
                ev.record()
                events[(g, rank)] = ev
                bar.wait(timeout=60)                 # host only: both events are recorded, nothing waits for the device
                torch.cuda.current_stream().wait_event(events[(g, 1 - rank)])
                torch.add(stage[(g & 1, 0)][:count], stage[(g & 1, 1)][:count], out=v)      # rank 0 + rank 1 on both ranks
                ev2 = torch.cuda.Event()
                ev2.record()
                events[("done", g, rank)] = ev2
                bar.wait(timeout=60)
                torch.cuda.current_stream().wait_event(events[("done", g, 1 - rank)])      # the peer has read my staging slot
            return hook

        def worker(rank):
            try:
                torch.cuda.set_device(0)
                first, cnt = (0, m // 2) if rank == 0 else (m // 2, m - m // 2)
                xs = torch.tensor(inp["xdata"][first:first + cnt], device=dev)
                ys = torch.tensor(inp["ydata"][first:first + cnt], device=dev)
                ws = None if inp["wdata"] is None else torch.tensor(inp["wdata"][first:first + cnt], device=dev)
                comm_len = int(capi.lib().splpak_plan_comm_len(nd, capi._p(np.ascontiguousarray(inp["nodes"], dtype=np.int32), capi._ip)))
                comm = torch.zeros(comm_len, dtype=torch.float64, device=dev)
                plan = capi.Plan(nd, inp["nodes"], inp["xmin"], inp["xmax"], inp["xtrap"], cnt, comm=comm)
                plan.set_allreduce(make_hook(rank, comm), rank, 2, stream_ordered=stream_ordered)
                coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
                stream = torch.cuda.Stream()
                outs = []
                for _ in range(2):
                    ierr, info = plan.fit(xs, ys, ws, coef, stream.cuda_stream)
                    stream.synchronize()
                    outs.append(coef.cpu().numpy().copy())
                plan.close()
                assert ierr == 0 and np.array_equal(outs[0], outs[1])
                results[rank] = outs[0]
            except Exception as exc:      # noqa: BLE001
                errors.append(exc)
                bar.abort()

        th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert not errors, errors
        return results

    ordered = run(True)
    synced = run(False)
    assert np.array_equal(ordered[0], ordered[1]) and np.array_equal(synced[0], synced[1])
    assert np.array_equal(ordered[0], synced[0])
    assert relmax(ordered[0], gold["coef"]) < 1e-10


@pytest.mark.gpu
def test_native_rccl_hook_one_rank_smoke(tmp_path, monkeypatch):
    """the library's own RCCL hook (csrc/rccl.hip: librccl opened at run time, ncclAllReduce on the fit's
    stream) through the C ABI -- no Python in the reductions.  This pool has ONE GPU, so the communicator has one rank;
    SPLPAK_RCCL_ONE_RANK_CALLS=1 makes the fit route its reductions (histogram, normal equations, every refinement residual)
    through ncclAllReduce anyway.  The communicator is made by the library from an id file, as a Fortran caller would."""
    from splpak_amd import capi
    from tests.cases import CASES, make_inputs
    monkeypatch.setenv("SPLPAK_RCCL_ONE_RANK_CALLS", "1")
    monkeypatch.setenv("SPLPAK_DEBUG_SUMS", "1")            # prints "before / after all-reduce" lines: the hook really ran
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = capi.rccl_comm_create_from_file(tmp_path / "splpak_nccl_id", 0, 1)
    assert comm
    try:
        for name in ("3d8", "2d16_zero_w"):
            inp = make_inputs(CASES[name])
            gold = load_golden(name)
            x = torch.tensor(inp["xdata"], device=dev)
            y = torch.tensor(inp["ydata"], device=dev)
            w = None if inp["wdata"] is None else torch.tensor(inp["wdata"], device=dev)
            coef = torch.zeros(int(np.prod(inp["nodes"])), dtype=torch.float64, device=dev)
            plan = capi.Plan(inp["ndim"], inp["nodes"], inp["xmin"], inp["xmax"], inp["xtrap"], x.shape[0])
            try:
                plan.set_rccl(comm, 0, 1)
                ierr, info = plan.fit(x, y, w, coef, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            finally:
                plan.close()
            assert ierr == 0 and relmax(coef.cpu().numpy(), gold["coef"]) < 1e-10 and info[9] < 1e-9
    finally:
        capi.rccl_comm_destroy(comm)
