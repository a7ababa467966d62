"""The N > 1 path on CPU: two ranks (gloo) each render their stripes of one frame through the same Context /
tile API the GPU path uses (the oracle stands in for the device), the tiles are gathered to rank 0 and
de-interleaved, and the result must be BIT-identical to a single full-frame render (seeds use global
coordinates, RaytracerCS.hlsl:430)."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import REPO, abi, scenes

tiling = importlib.import_module("gpu-raytracer_amd.tiling")


def test_owned_rows_partition_every_row_once():
    for H, sr, world in [(1080, 8, 8), (1080, 8, 3), (37, 5, 4), (16, 16, 1), (9, 4, 4)]:
        allrows = np.concatenate([tiling.owned_rows(H, sr, r, world) for r in range(world)])
        assert sorted(allrows) == list(range(H))
        assert tiling.max_owned_rows(H, sr, world) >= H // world


def test_tile_api_matches_owned_rows(oracle):
    ctx = oracle.context()
    ctx.resize(20, 37)
    for sr, world in [(5, 4), (8, 2), (37, 1), (1, 3)]:
        for r in range(world):
            ctx.set_tile(sr, r, world)
            assert ctx.tile_rows() == len(tiling.owned_rows(37, sr, r, world))
    ctx.close()


def _rank_main(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    import conftest
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = conftest.Oracle()
    desc = conftest.scenes.small_mesh_scene(seed=2, n_objects=4, detail=5, with_lights=True)
    W, H, SR, FRAMES = 48, 27, 4, 3
    rt, sc, res = oracle.instantiate(desc, W, H)
    rt.ctx.set_tile(SR, rank, world)
    for _ in range(FRAMES):
        rt.render()
    tile = rt.ctx.read_accum()
    pad = tiling.max_owned_rows(H, SR, world)
    buf = torch.zeros((pad, W, 4), dtype=torch.float32)
    buf[: tile.shape[0]] = torch.from_numpy(tile)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    # samples split (bench.py --split samples): same pixels, disjoint frame indices, one SUM reduce
    rt.ctx.set_tile(H, 0, 1)
    rt.resetAccumulation()
    rt.render_data.numAccumulationFrames = rank * 2  # rank r renders frames 2r+1, 2r+2
    rt.renderFrames(2)
    part = torch.from_numpy(rt.ctx.read_accum())
    dist.reduce(part, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        full = tiling.deinterleave(gathered, H, SR).numpy()
        ref_rt, _, _ = oracle.instantiate(desc, W, H)
        for _ in range(FRAMES):
            ref_rt.render()
        ref = ref_rt.ctx.read_accum()
        ref_rt.render()  # 4 frames = what the two ranks' sample blocks add up to
        ref4 = ref_rt.ctx.read_accum()
        samples_ok = bool(np.allclose(part.numpy(), ref4, rtol=1e-5, atol=1e-6, equal_nan=True) and (part.numpy()[..., 3] == 2 * world).all())
        np.save(out_path, np.array([np.array_equal(full, ref, equal_nan=True) and samples_ok, float(np.nanmax(np.abs(full - ref)))]))
    dist.barrier()
    dist.destroy_process_group()


def _rank_4k(rank, world, port, out_path, stripe):
    """Config 5's partition at its real height (2160 rows; narrow so that CPU ranks stay cheap): `world` gloo ranks, each
    renders the stripes okrt_set_tile(stripe, rank, world) gives it, tiles padded to the largest and gathered to rank 0."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    import conftest
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = conftest.Oracle()
    desc = conftest.scenes.small_mesh_scene(seed=5, n_objects=2, detail=4, with_textures=False)
    W, H = 24, 2160
    rt, sc, res = oracle.instantiate(desc, W, H)
    oracle.lib.okrt_oracle_set_threads(rt.ctx.h, 1)
    rt.ctx.set_tile(stripe, rank, world)
    assert rt.ctx.tile_rows() == len(tiling.owned_rows(H, stripe, rank, world))
    rt.renderFrames(2)
    tile = rt.ctx.read_accum()
    pad = tiling.max_owned_rows(H, stripe, world)
    buf = torch.zeros((pad, W, 4), dtype=torch.float32)
    buf[: tile.shape[0]] = torch.from_numpy(tile)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank == 0:
        full = tiling.deinterleave(gathered, H, stripe).numpy()
        rt.ctx.set_tile(H, 0, 1)
        rt.resetAccumulation()
        oracle.lib.okrt_oracle_set_threads(rt.ctx.h, 4)
        rt.renderFrames(2)
        ref = rt.ctx.read_accum()
        np.save(out_path, np.array([float(np.array_equal(full, ref, equal_nan=True) and (full[..., 3] == 2).all()), float(np.nanmax(np.abs(full - ref)))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,stripe", [(4, 8), (8, 8), (8, 270)])
def test_4k_height_splits_over_4_and_8_ranks(tmp_path, world, stripe):
    """4 and 8 gloo ranks on the 2160-row frame: interleaved 8-row stripes (bench.py's split) and 270-row bands (SURVEY
    8e v1). What the 8-GPU run adds is only the RCCL transport under the same gather."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    port = 31500 + (os.getpid() % 2000) + world + stripe
    mp.spawn(_rank_4k, args=(world, port, out, stripe), nprocs=world, join=True)
    ok, maxdiff = np.load(out)
    assert ok == 1.0, maxdiff


def test_two_rank_tile_split_equals_full_frame(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, out), nprocs=2, join=True)
    ok, maxdiff = np.load(out)
    assert ok == 1.0, maxdiff
