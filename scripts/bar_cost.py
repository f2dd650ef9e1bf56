"""Same-box A/B of the progress bar's postfix update: forced refresh every iteration against the bar's own interval."""
import os, sys, time
import torch, tqdm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.loadest_gp import LoadestGP
from tests.helpers import loadest_dataset
ld = loadest_dataset(300, seed=1)
m = LoadestGP(); m.fit(*ld, iterations=3)
orig = tqdm.tqdm.set_postfix_str
def forced(self, s="", refresh=True):
    return orig(self, s, refresh=True)
for label in ("interval", "forced", "interval", "forced"):
    tqdm.tqdm.set_postfix_str = orig if label == "interval" else forced
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(*ld, iterations=150); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 150 * 1e3)
    print(f"postfix refresh {label:8s}: {best:.3f} ms/iter")
