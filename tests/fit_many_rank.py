"""One rank of the `fit_many_distributed` rehearsal on the GPU box (started by tests/test_gpu_fit_many_dist.py through
torch.distributed.run; the ranks share the one GPU over gloo -- RCCL refuses two ranks on one device).
argv: family, n_sites, iterations, output directory."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sites(family, count):
    from tests.helpers import loadest_dataset, rating_dataset

    if family == "loadest":
        from discontinuum_amd.loadest_gp import LoadestGP

        return [LoadestGP() for _ in range(count)], [loadest_dataset(150 + 37 * i, seed=30 + i) for i in range(count)]
    from discontinuum_amd.rating_gp import RatingGP

    return [RatingGP() for _ in range(count)], [rating_dataset(140 + 29 * i, seed=50 + i) for i in range(count)]


def flat(m):
    return torch.cat([p.detach().reshape(-1).double() for _, p in sorted(m.model.named_parameters())]
                     + [p.detach().reshape(-1).double() for _, p in sorted(m.likelihood.named_parameters())]).numpy()


def main():
    family, count, iters, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("DGP_BENCH_BACKEND", "gloo"))
    rank = dist.get_rank()
    from discontinuum_amd.multisite_fit import fit_many_distributed

    models, data = sites(family, count)
    objs, table = fit_many_distributed(models, data, iterations=iters)
    other = (rank + 1) % count  # with two ranks: a site this rank did not train
    mu, se = models[other].predict(data[other][0])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), objs=objs.numpy(), table=table.numpy(),
             params=np.concatenate([flat(m) for m in models]), pred=np.asarray(mu.values), other=other)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
