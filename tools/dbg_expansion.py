import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyFunction
dev = torch.device("cuda:0")
for P, n in ((512, 1024), (256, 512), (64, 256), (16, 64)):
    rng = np.random.default_rng(P)
    x = rng.random((1, n, 3), dtype=np.float32)
    d0, a0, m0 = oracle.expansion_forward(x, P, 1.5)
    d, a, m = expansionPenaltyFunction.apply(torch.from_numpy(x).to(dev), P, 1.5)
    d, a, m = d.cpu().numpy(), a.cpu().numpy(), m.cpu().numpy()
    bad = np.nonzero(d != d0)
    print(P, "assign eq", np.array_equal(a, a0), "dist mismatches", len(bad[0]), "mean", m, m0 / np.float32(n / P))
    for i in range(min(5, len(bad[0]))):
        j = bad[1][i]
        print("   j", j, "hip", d[0, j], "oracle", d0[0, j], "assign", a[0, j], a0[0, j],
              "true", np.linalg.norm(x[0, j].astype(np.float64) - x[0, a0[0, j]]))
