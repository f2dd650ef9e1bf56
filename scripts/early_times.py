"""Stage times of one fit step (HIP events inside the library) with the early inverse on (level 2) and off (level 1)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
from discontinuum_amd import _lib
from oracle.gp_oracle import synth_loadest
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0"); dt = torch.float64
X, y = synth_loadest(n, 3, 0)
X = torch.tensor(X, dtype=dt, device=dev).contiguous(); y = torch.tensor(y, dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931471805599453] * 11
for level in (1, 2):
    p = GPPlan("loadest", n, 3, dtype=dt, device=dev, lookahead=level)
    p.set_inputs(X); p.set_timing(True)
    for _ in range(3): p.fit_step(theta, y, noise)
    torch.cuda.synchronize()
    acc = np.zeros(_lib.TIME_COUNT)
    for _ in range(5):
        p.fit_step(theta, y, noise); torch.cuda.synchronize(); acc += np.array(p.get_timing())
    acc /= 5
    names = ["gram", "potrf", "trtri", "lauum", "solve", "grad"]
    print(f"level {level}: " + "  ".join(f"{k} {acc[getattr(_lib, 'TIME_' + k.upper())]:.3f}" for k in names) +
          f"  | sum {sum(acc[getattr(_lib, 'TIME_' + k.upper())] for k in names):.3f} ms")
