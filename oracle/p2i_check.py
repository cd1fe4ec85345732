"""Winner-id comparison for the p2i max splat: ids must equal the oracle's on EVERY pixel, except where the two
candidates' exact values are within one fp32 ulp of each other -- there the HIP path (fp64 series instead of
glibc's cos) and the oracle may legitimately order them differently.  The exceptions are checked one by one
with the oracle's arithmetic (oracle/p2i.c: fp32 distance, double cosine, narrowed) and counted."""
import numpy as np


def _value(pid, c, y, x, pts, feat, R):
    py, px = pts[pid, 0], pts[pid, 1]
    dx = x.astype(np.float32) - px
    dy = y.astype(np.float32) - py
    r = np.sqrt(dx * dx + dy * dy, dtype=np.float32)
    w = (np.cos(r.astype(np.float64) * np.pi / np.float64(np.float32(R))) * 0.5 + 0.5).astype(np.float32)
    return feat[pid, c] * w, r


def assert_ids_exact_up_to_ulp_ties(ids, ref_ids, pts, feat, background, R, what=""):
    """ids / ref_ids [B, C, H, W]; returns the number of (verified) one-ulp ties."""
    ids, ref_ids = np.asarray(ids), np.asarray(ref_ids)
    bad = np.argwhere(ids != ref_ids)
    if len(bad) == 0:
        return 0
    b, c, y, x = bad.T
    ph, pr = ids[b, c, y, x], ref_ids[b, c, y, x]
    pts, feat = np.asarray(pts, np.float32), np.asarray(feat, np.float32)
    bgv = np.broadcast_to(np.asarray(background, np.float32), ids.shape)[b, c, y, x]
    vals = []
    for p in (ph, pr):
        v, r = _value(np.maximum(p, 0), c, y, x, pts, feat, R)
        assert np.all((p < 0) | (r <= np.float32(R))), f"{what}: a winner id outside its kernel radius"
        vals.append(np.where(p < 0, bgv, v).astype(np.float32))
    vh, vr = vals
    ulp = np.spacing(np.maximum(np.abs(vh), np.abs(vr)).astype(np.float32))
    worst = np.abs(vh.astype(np.float64) - vr.astype(np.float64)) / ulp
    assert np.all(worst <= 1.0), (f"{what}: {int((worst > 1).sum())} of {len(bad)} differing winner ids are NOT "
                                  f"one-ulp ties (worst {worst.max():.1f} ulp)")
    return len(bad)
