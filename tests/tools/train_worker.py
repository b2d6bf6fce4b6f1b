"""Launched by tests/test_gpu_reference_flow.py under `python -m torch.distributed.run --nproc-per-node 2` with
NERFMESHES_RANKS_PER_GPU=2 (two ranks on the one GPU, gloo): every rank runs `nerfmeshes_amd.train_nerf.main(argv)` -- the
reference's train_nerf.py command line -- WITHOUT --deterministic (each rank builds its model from its own unseeded generator,
as the reference does), then the ranks compare what they hold: parameters equal on every rank (rank 0's initial weights were
broadcast, gradients averaged every step), the rays they trained on different, ONE version directory.  Prints TRAIN_DDP_OK."""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import dist as nd, train_nerf  # noqa: E402


def main():
    cfg_path, deterministic = sys.argv[1], sys.argv[2] == "1"
    argv = ["--config", cfg_path, "--run-name", "ddp", "--gpus", os.environ["WORLD_SIZE"]] + (["--deterministic"] if deterministic else [])
    seen = []
    from nerfmeshes_amd.models import NeRFModel
    step = NeRFModel.training_step

    def spy(self, batch, idx):                      # the rays this rank trains on
        seen.append(batch["ray_directions"].detach().float().sum().reshape(1).cpu())
        return step(self, batch, idx)

    NeRFModel.training_step = spy
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer, model, pp = train_nerf.main(argv)
    rank, world = nd.world()
    dev = next(model.parameters()).device
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    copies = nd.all_gather_rows(flat[None].contiguous(), [1] * world)
    assert all(torch.equal(copies[0], copies[r]) for r in range(world)), "replicas hold different parameters after training"
    rays = nd.all_gather_rows(torch.cat(seen).to(dev)[None].contiguous(), [1] * world)
    assert not torch.equal(rays[0], rays[-1]), "both ranks trained on the same rays: data parallelism was a no-op"
    versions = sorted(os.listdir(os.path.dirname(str(pp.log_dir))))
    assert versions == ["version_0"], versions
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        ck = torch.load(os.path.join(str(pp.log_dir), "checkpoints", "model_last.ckpt"), weights_only=False)
        saved = torch.cat([ck["state_dict"][k].reshape(-1) for k, _ in model.named_parameters()]).to(dev)
        assert torch.equal(saved, flat), "the checkpoint is not the replicas' model"
        print(f"TRAIN_DDP_OK world={world} steps={ck['global_step']} params={flat.numel()}", flush=True)
    nd.shutdown()


if __name__ == "__main__":
    main()
