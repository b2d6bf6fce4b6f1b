"""CPU, world_size 2, gloo: the multi-GPU sharding layer (nerfmeshes_amd.dist) assembles exactly what a single
process produces -- rays / grid slabs are disjoint contiguous ranges, one all-gather at the end."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfmeshes_amd import dist as nd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def _fake_pixels(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float32)
    return torch.stack([idx, idx * 0.5, idx + 1000.0], -1)


def _render_job(rank, world):
    full = nd.render_view_sharded(_fake_pixels, 1001)          # ragged: 501 + 500 rays
    even = nd.render_view_sharded(_fake_pixels, 640)           # equal shards: single-collective fast path
    return full.numpy(), even.numpy()


def _grid_job(rank, world):
    n0, n1, n2 = 7, 3, 5

    def query(lo, hi):
        return torch.arange(lo * n1 * n2, hi * n1 * n2, dtype=torch.float32) * 2.0

    return nd.density_grid_sharded(query, n0, n1, n2).numpy()


def test_split_range_is_a_partition():
    for n in (0, 1, 7, 640000, 480):
        for w in (1, 2, 3, 8):
            parts = [nd.split_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n0", [2, 3, 7, 33, 480])
@pytest.mark.parametrize("ws", [1, 2, 3, 8, 11])
def test_slab_layers_cover_the_cube_layers_once_with_the_right_ghosts(n0, ws):
    """Per-slab marching cubes (dist.slab_layers): the n0 - 1 cube layers are dealt out exactly once, in rank order; a
    non-empty slab carries a ghost layer below unless it starts at layer 0 and one above unless it ends at the top, and
    asks for exactly the voxel planes those layers touch; ranks beyond the layer count get empty slabs and no planes'
    worth of work.  (The kernels behind it need a GPU: tests/test_gpu_mc.py, tests/test_gpu_dist.py.)"""
    covered = []
    for r in range(ws):
        lo, hi, below, above, p_lo, p_hi = nd.slab_layers(n0, r, ws)
        assert 0 <= lo <= hi <= n0 - 1
        covered += list(range(lo, hi))
        if hi == lo:
            assert below == 0 and above == 0
            continue
        assert below == int(lo > 0) and above == int(hi < n0 - 1)
        assert (p_lo, p_hi) == (lo - below, hi + 1 + above) and 0 <= p_lo and p_hi <= n0
        # a slab with ghosts still has a cube layer of its own (nm_mc_count_slab's precondition)
        assert (p_hi - p_lo) - 1 - below - above == hi - lo >= 1
    assert covered == list(range(n0 - 1))


def test_single_process_is_identity():
    assert nd.world() == (0, 1)
    out = nd.render_view_sharded(_fake_pixels, 10)
    assert torch.equal(out, _fake_pixels(0, 10))


@pytest.mark.timeout(120)
def test_render_view_sharded_two_ranks_equals_single():
    res = _run(_render_job)
    for full, even in res:
        assert np.array_equal(full, _fake_pixels(0, 1001).numpy())
        assert np.array_equal(even, _fake_pixels(0, 640).numpy())


@pytest.mark.timeout(120)
def test_density_grid_sharded_two_ranks_equals_single():
    res = _run(_grid_job)
    ref = (torch.arange(7 * 3 * 5, dtype=torch.float32) * 2.0).view(7, 3, 5).numpy()
    for grid in res:
        assert np.array_equal(grid, ref)


def _grad_job(rank, world):
    torch.manual_seed(0)                                        # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.zeros(4))                 # a parameter without a gradient on rank 0
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) * (rank + 1) / 10.0
    loss = net(x).pow(2).mean() + (unused.sum() if rank == 1 else 0.0)
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    nd.all_reduce_gradients(list(net.parameters()) + [unused], bucket_bytes=64)    # several buckets
    return [g.numpy() for g in local], [p.grad.numpy() for p in net.parameters()], unused.grad.numpy()


def test_gradient_all_reduce_two_ranks_is_the_mean():
    res = _run(_grad_job)
    for k in range(4):
        mean = 0.5 * (res[0][0][k] + res[1][0][k])
        for r in range(2):
            np.testing.assert_allclose(res[r][1][k], mean, rtol=1e-6, atol=1e-7)
    for r in range(2):
        np.testing.assert_allclose(res[r][2], np.full(4, 0.5, dtype=np.float32))


# ---- eval_nerf's per-view loss gather: unequal and empty shards (views % world != 0) -------------------------------
def _fake_render(model, pose, h, w, focal, bounds, chunksize, device):      # rgb depends on the view only
    v = float(pose[0, 0])
    rgb = torch.full((h * w, 3), 0.25) + 0.01 * v * torch.arange(h * w * 3, dtype=torch.float32).reshape(-1, 3) / (h * w * 3)
    return rgb, rgb[:, 0]


def _eval_job(rank, world, num_views=1):
    from nerfmeshes_amd import eval_nerf as E
    from nerfmeshes_amd import synthetic as S
    from nerfmeshes_amd.models.model_helpers import nest_dict
    from nerfmeshes_amd.nerf import CfgNode
    E.render_view = _fake_render
    views = []
    for i in range(num_views):
        pose = np.eye(4, dtype=np.float32) * (i + 1)
        tgt = None if (num_views == 5 and i == 1) else torch.full((6 * 4, 3), 0.5)      # one view without targets
        views.append((pose, 6, 4, 10.0, tgt))
    cfg = CfgNode(nest_dict(S.hparams(chunksize=8), sep="."))
    losses, total, psnr, _ = E.eval_views(None, views, cfg, device="cpu")
    return [float(x) for x in losses], None if total is None else float(total)


@pytest.mark.timeout(180)
@pytest.mark.parametrize("num_views", [1, 3, 4, 5])
def test_eval_loss_gather_ragged_views(num_views):
    """1 view on 2 ranks (rank 1 holds none), 3 and 5 views (unequal counts; one view unscored):
    every rank must enter the collective and get the losses back in VIEW order."""
    import functools
    single = _eval_job(0, 1, num_views)              # no process group: plain loop
    res = _run(functools.partial(_eval_job, num_views=num_views))
    for losses, total in res:
        assert losses == pytest.approx(single[0], rel=0, abs=0)
        assert total == pytest.approx(single[1], rel=1e-7)


def test_init_from_env_gloo(monkeypatch):
    """`init_from_env` reads the launcher variables, joins the group and is a no-op without them."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert nd.init_from_env()[:2] == (0, 1)
    monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    try:
        rank, ws, dev = nd.init_from_env(backend="gloo")
        assert (rank, ws) == (0, 1) and dist.is_initialized()
        # an initialised one-rank group still runs the collective
        out = nd.all_gather_rows(torch.arange(6.0).reshape(3, 2), [3])
        assert torch.equal(out, torch.arange(6.0).reshape(3, 2))
        with pytest.raises(ValueError):
            nd.all_gather_rows(torch.zeros(2, 2), [3])
    finally:
        nd.shutdown()
    assert not dist.is_initialized()


# ---- all-gatherv: exact-size ragged gather, empty shards included ------------------------------
def _ragged_job(rank, world):
    counts = [5, 0, 2][:world] if world == 3 else [0, 4]
    rows = torch.arange(counts[rank] * 3, dtype=torch.float32).reshape(counts[rank], 3) + 100.0 * rank
    ids = torch.arange(counts[rank], dtype=torch.int32) + 1000 * rank
    return nd.all_gather_rows(rows, counts).numpy(), nd.all_gather_ragged(ids).numpy()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_v_places_ragged_and_empty_shards_exactly(world):
    counts = [5, 0, 2] if world == 3 else [0, 4]
    want_rows = np.concatenate([np.arange(c * 3, dtype=np.float32).reshape(c, 3) + 100.0 * r for r, c in enumerate(counts)])
    want_ids = np.concatenate([np.arange(c, dtype=np.int32) + 1000 * r for r, c in enumerate(counts)])
    for rows, ids in _run(_ragged_job, world=world):
        assert rows.shape == (sum(counts), 3), "exact size: no padding to the largest shard"
        assert np.array_equal(rows, want_rows) and np.array_equal(ids, want_ids)


def _count_job(rank, world):
    """Collectives a ragged gather costs: ONE for known counts, TWO when the counts travel first; the oversized-payload
    fallback (one broadcast per non-empty rank) gives the same rows; NERFMESHES_DIST_DEBUG catches ranks that disagree."""
    counts = [5, 0, 2][:world] if world == 3 else [3, 4]
    rows = torch.arange(counts[rank] * 3, dtype=torch.float32).reshape(counts[rank], 3) + 100.0 * rank
    c0 = nd.COLLECTIVES["count"]
    a = nd.all_gather_rows(rows, counts)
    c1 = nd.COLLECTIVES["count"]
    b = nd.all_gather_ragged(rows)
    c2 = nd.COLLECTIVES["count"]
    limit, nd.PADDED_GATHER_LIMIT = nd.PADDED_GATHER_LIMIT, 0
    c = nd.all_gather_rows(rows, counts)
    nd.PADDED_GATHER_LIMIT = limit
    c3 = nd.COLLECTIVES["count"]
    os.environ["NERFMESHES_DIST_DEBUG"] = "1"
    nd.all_gather_rows(rows, counts)                      # agreeing counts pass the check
    wrong = list(counts)
    if rank == 1:
        wrong[0] += 1                                      # rank 1 believes rank 0 holds one row more
    try:
        nd.all_gather_rows(rows, wrong)
        caught = False
    except RuntimeError as e:
        caught = "disagree" in str(e)
    os.environ.pop("NERFMESHES_DIST_DEBUG")
    return (c1 - c0, c2 - c1, c3 - c2, torch.equal(a, b) and torch.equal(a, c), caught)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_ragged_gathers_are_single_collectives(world):
    nonempty = 2
    for one, two, fallback, same, caught in _run(_count_job, world=world):
        assert (one, two, fallback) == (1, 2, nonempty), (one, two, fallback)
        assert same and caught


# ---- the Trainer stand-in under a launcher: identical replicas, distinct ray streams, one version dir ----
def _assemble_job_7(rank, world):
    return _assemble_job(rank, world, 7)


def _assemble_job_3(rank, world):
    return _assemble_job(rank, world, 3)


def _assemble_job(rank, world, n0):
    """What `mesh_nerf --gather triangles --use-cached-mesh` does for the mesh cache: every rank holds the planes of its slab
    (ghost planes included) and the whole grid is assembled from the planes each rank accounts for."""
    from nerfmeshes_amd import mesh_nerf
    n1, n2 = 3, 5
    full = torch.arange(n0 * n1 * n2, dtype=torch.float32).reshape(n0, n1, n2) * 0.5
    lo, hi, below, above, p_lo, p_hi = nd.slab_layers(n0, rank, world)
    slab = None if hi == lo else full[p_lo:p_hi].reshape(-1).clone()        # a rank without a cube layer holds no slab
    return mesh_nerf._assemble_grid_from_slabs(slab, (n0, n1, n2), torch.device("cpu")).numpy()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,job,n0", [(2, _assemble_job_7, 7), (3, _assemble_job_7, 7), (4, _assemble_job_3, 3)])
def test_mesh_cache_grid_is_assembled_from_the_ranks_slabs(world, job, n0):
    """mesh_nerf.py:139-144 under --gather triangles: the cached density grid equals the single-process grid on every rank --
    ghost planes counted once, the top plane by the last rank that has a cube layer, ranks without a layer (4 ranks, 2 layers)
    contributing nothing but entering the collective."""
    want = (np.arange(n0 * 3 * 5, dtype=np.float32).reshape(n0, 3, 5) * 0.5)
    for got in _run(job, world=world):
        assert got.shape == want.shape and np.array_equal(got, want)


def _fit_worker(rank, world, port, tmp, deterministic, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from nerfmeshes_amd import lightning_compat as LC
    nd.init_from_env(backend="gloo")
    try:
        if deterministic:
            LC.seed_everything(42)          # train_nerf.py --deterministic: every rank seeds alike

        class Toy(LC.LightningModule):
            def __init__(self):
                super().__init__()
                self.hparams = {"toy": 1}
                self.net = torch.nn.Linear(4, 3)          # unseeded unless --deterministic: differs per rank
                self.register_buffer("stat", torch.rand(2))
                self.seen = []

            def setup(self, stage):
                self.trainer.max_steps = 6

            def configure_optimizers(self):
                return torch.optim.SGD(self.parameters(), lr=0.1)

            def train_dataloader(self):
                return [torch.zeros(1) for _ in range(6)]

            def training_step(self, batch, batch_idx):
                x = torch.rand(8, 4)                      # the "random rays" of this rank
                self.seen.append(x.clone())
                return {"loss": self.net(x).pow(2).mean(), "log": {}}

        logger = LC.TensorBoardLogger(tmp, "run")
        try:                                              # a bare property read must not communicate: it raises instead
            logger.version
            raise AssertionError("version read before resolve_version() must raise in a multi-rank job")
        except RuntimeError as e:
            assert "resolve_version" in str(e)
        logger.resolve_version()                          # what PathParser.parse does, on every rank ...
        os.makedirs(os.path.join(logger.log_dir, "checkpoints"), exist_ok=True)      # ... before it creates the directory
        model = Toy()
        first = [p.detach().clone() for p in model.parameters()]
        LC.Trainer(logger=logger, max_epochs=1).fit(model)
        out[rank] = {"version": logger.version, "first": [p.numpy() for p in first], "stat": model.stat.numpy(),
                     "params": [p.detach().numpy() for p in model.parameters()], "rays": torch.stack(model.seen).numpy()}
    finally:
        nd.shutdown()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("deterministic", [False, True])
def test_trainer_fit_keeps_two_replicas_identical_on_different_rays(tmp_path, deterministic):
    """`train_nerf` under `torch.distributed.run`: rank 0's initial weights and buffers reach every rank (DDP's broadcast),
    so gradient averaging keeps the replicas equal step after step; the ranks draw DIFFERENT rays even when
    `--deterministic` seeded them alike; and all ranks log into the version directory rank 0 picked."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path), deterministic, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["version"] == b["version"] == 0
    assert sorted(os.listdir(tmp_path / "run")) == ["version_0"]
    if not deterministic:
        assert any(not np.array_equal(x, y) for x, y in zip(a["first"], b["first"])), "the test needs replicas that start apart"
    for x, y in zip(a["params"], b["params"]):
        assert np.array_equal(x, y), "replicas diverged"
    assert np.array_equal(a["stat"], b["stat"])
    assert any(not np.array_equal(x, y) for x, y in zip(a["params"], a["first"])), "no training happened"
    assert not np.array_equal(a["rays"], b["rays"]), "both ranks drew the same rays: data parallelism would be a no-op"
