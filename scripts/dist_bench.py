"""BASELINE config 5 across the GPUs of a node: NLL of ONE n = 65536 fp32 matrix, block-cyclic over the ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 scripts/dist_bench.py [n] [W]

One rank per GPU over RCCL; with DGP_BENCH_BACKEND=gloo several ranks may share one GPU (functional rehearsal only).
Prints the wall time of `distributed_nll` (Gram build on every rank + distributed factorisation + forward solve)."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
from discontinuum_amd.dist_chol import distributed_nll
from oracle.gp_oracle import synth_loadest

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
dev = torch.device("cuda", local % max(1, torch.cuda.device_count())); torch.cuda.set_device(dev)
if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("DGP_BENCH_BACKEND", "nccl")
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
dt = torch.float32
X, y = synth_loadest(n, 3, 0)
p = GPPlan("loadest", n, 3, dtype=dt, device=dev); p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
yd = torch.tensor(y, dtype=dt, device=dev); noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931471805599453] * 11
out = distributed_nll(p, theta, yd, noise, group_panels=W); torch.cuda.synchronize()
if world > 1: dist.barrier()
t0 = time.perf_counter(); reps = 2
for _ in range(reps): out = distributed_nll(p, theta, yd, noise, group_panels=W)
torch.cuda.synchronize()
if world > 1: dist.barrier()
ms = (time.perf_counter() - t0) / reps * 1e3
if rank == 0:
    print(f"n={n} fp32, {world} rank(s), groups of {W} panels: {ms:.1f} ms per NLL  ({n**3 / 3 / ms / 1e9:.1f} TFLOP/s aggregate)  "
          f"NLL={float(out[0]):.3f} info={int(out[3])}")
if world > 1: dist.destroy_process_group()
