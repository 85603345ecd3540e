"""Shared parity checks of the -m gpu tests (TEST INFRASTRUCTURE: compares the HIP path with the CPU oracle).

`assert_codes_near_tie`: the rule SURVEY.md section 7 states for END-TO-END RVQ codes (wav -> SEANet encoder -> RVQ).  On
IDENTICAL latents the codes are bit exact (tested separately); end to end the device's latents differ from the oracle's
by fp32 round-off `delta`, and an index may differ ONLY where the oracle's own decision was a near tie that `delta` can
flip.  With r the oracle's residual at the first level where a frame's codes differ, c* the oracle's choice and c the
device's,

    dist(c; r + delta) <= dist(c*; r + delta)   <=>   dist(c; r) - dist(c*; r) <= 2 delta . (e_c - e_c*)

(dist(c; r) = |r - e_c|^2, core_vq.py:164-172), so the oracle's margin `dist(c; r) - dist(c*; r)` (>= 0) must not exceed
`2 |delta| |e_c - e_c*|` plus the fp32 rounding of the two distance evaluations.  Levels below a frame's first flip see
a different residual and are reported as "downstream", not judged.
"""
import torch


def assert_codes_near_tie(codes_dev: torch.Tensor, codes_ref: torch.Tensor, lat_dev: torch.Tensor, lat_ref: torch.Tensor,
                          codebooks: torch.Tensor, what: str = '', fp_slack: float = 4e-6):
    """codes_dev [B, K, T] int64 (device, end to end), codes_ref the oracle's codes of lat_ref, lat_dev / lat_ref [B, D, T] f32
    (device / oracle encoder output), codebooks [K, bins, D] f32.  Asserts the near-tie rule per RVQ level and prints the
    per-level report; returns the number of frames with a flipped level."""
    codes_dev, codes_ref, lat_dev, lat_ref, codebooks = (t.detach().cpu() for t in (codes_dev, codes_ref, lat_dev, lat_ref, codebooks))
    assert codes_dev.shape == codes_ref.shape, (codes_dev.shape, codes_ref.shape)
    B, K, T = codes_dev.shape
    D = lat_ref.shape[1]
    r = lat_ref.permute(0, 2, 1).reshape(B * T, D).double()
    delta = (lat_dev.double() - lat_ref.double()).permute(0, 2, 1).reshape(B * T, D)
    dn = delta.norm(dim=1)
    alive = torch.ones(B * T, dtype=torch.bool)          # frames whose codes agreed on every level so far
    report, worst = [], 0.0
    for k in range(K):
        e = codebooks[k].double()                         # [bins, D]
        dist = (r * r).sum(1, keepdim=True) - 2.0 * r @ e.t() + (e * e).sum(1)[None]
        c_ref = codes_ref[:, k].reshape(B * T)   # the oracle's own fp32 decision (its f64 margin may be ~ -1 ulp)
        c_dev = codes_dev[:, k].reshape(B * T)
        flip = alive & (c_dev != c_ref)
        n_down = int((~alive).sum())
        agree = float((c_dev == c_ref).float().mean())
        if flip.any():
            i = flip.nonzero().squeeze(1)
            margin = dist[i, c_dev[i]] - dist[i, c_ref[i]]
            e_gap = (e[c_dev[i]] - e[c_ref[i]]).norm(dim=1)
            bound = 2.0 * dn[i] * e_gap + fp_slack * (1.0 + dist[i, c_ref[i]].abs() + (r[i] * r[i]).sum(1))
            ratio = float((margin / bound).max())
            worst = max(worst, ratio)
            report.append(f"L{k}: {int(flip.sum())} first flips (max margin {float(margin.max()):.2e}, "
                          f"margin/bound {ratio:.2f}), {n_down} downstream, agree {agree:.4f}")
            assert ratio <= 1.0, (f"{what}: RVQ level {k}: a differing code is NOT a near tie "
                                  f"(margin {float(margin.max()):.3e} > bound; margin/bound {ratio:.2f})")
        else:
            report.append(f"L{k}: 0 first flips, {n_down} downstream, agree {agree:.4f}")
        alive = alive & ~flip
        r = r - e[c_ref]
    total_flips = int((~alive).sum())
    print(f"[near-tie] {what}: {B * T} frames x {K} levels, |delta|_max {float(delta.abs().max()):.2e}, "
          f"{total_flips} frames with a flipped level (all near ties, worst margin/bound {worst:.2f})")
    for line in report[:8] + (['...'] if len(report) > 8 else []):
        print(f"[near-tie]   {line}")
    return total_flips
