"""Score-folded cross-attention tables of the decode step (include/acmi.h, acmi_lm_layer.w_qkvs / w_g2 / xs_u).

The reference's cross-attention block (transformer.py:344-361, 563-566) is, per decode position and row b,

    q   = norm_cross(x1) W_q^T + b_q                    x1 = x0 + att W_out^T (+ b_out)
    s   = scale * q_h . K[b, h, j]                       (H heads x Lc source positions)
    p   = softmax_j(s)
    x2  = x1 + (sum_j p[h, j] V[b, h, j]) W_cout^T (+ b_cout)

K and V are constant over a generate (projected once), so both contractions over the head dimension can be moved into
per-generate tables -- the same algebra that already gives the cross-attention query no launch of its own (w_qkvx / w_mq):

    G[b, hj, :]  = scale * sum_{f in h} K[b, h, j, f] W_q'[f, :]          W_q' = W_q diag(gamma_cross)      [R, H Lc, d]
    G2           = G W_out                                                 (the att part of x1)
    U[b, hj, :]  = sum_{f in h} V[b, h, j, f] W_cout[:, f]                                                   [R, H Lc, d]

    S_raw[b]     = (x0[b] - shift[b]) G[b]^T  +  att[b] G2[b]^T (+ G[b] b_out)     rides in the QKV / out-projection launches
    s[b]         = rstd[b] (S_raw[b] - (mean[b] - shift[b]) CS[b]) + BS[b]          the folded LayerNorm, CS = row sums of G
    x2[b]        = x1[b] + p[b] U[b] (+ b_cout)                                      ONE launch instead of attention + GEMM

The scores leave the GEMM launches finished, so the launch that replaces cross-attention AND its output projection reads
1.5 KB of fresh scores per row instead of the query and the keys, and streams U (R H Lc d elements) instead of W_cout -- one
dependent launch less per layer.  Worth it while R H Lc <= 2 d (MusicGen-medium, 8 conditioned rows x 24 heads x 16 text
positions = 2 d); the caller falls back to the separate launches otherwise.

Everything here is torch tensor algebra on the device the inputs live on (a once-per-generate weight fold, like the
LayerNorm / LayerScale folds of LMModel._pack); tests/test_host_cpu.py pins it against the direct computation on the CPU.
"""
import typing as tp

import torch


def fold_tables(kc: torch.Tensor, vc: torch.Tensor, wq_folded: torch.Tensor, b_q: torch.Tensor, w_out: torch.Tensor,
                b_out: tp.Optional[torch.Tensor], w_cout: torch.Tensor, wdtype: torch.dtype):
    """kc, vc [R, H, Lc, hd]: the cross-attention caches of the conditioned rows as f32 VALUES of the cache's element type.
    wq_folded [d, d] = W_q diag(gamma_cross), b_q [d] = W_q beta_cross (+ the projection's own bias), w_out / w_cout [d, d]
    (LayerScale already multiplied in), b_out [d] or None; all f32.  Returns a dict of
        G, G2, U   [R, H Lc, d] rounded to `wdtype`        CS, BS [R, H Lc] f32        b_gs [R, H Lc] f32 or None."""
    R, H, Lc, hd = kc.shape
    d = wq_folded.shape[0]
    assert H * hd == d and vc.shape == kc.shape
    scale = float(hd) ** -0.5
    wq_h = wq_folded.view(H, hd, d)                                   # rows of W_q' grouped by head
    G = torch.einsum('bhjf,hfk->bhjk', kc, wq_h).mul_(scale).reshape(R, H * Lc, d)
    BS = (torch.einsum('bhjf,hf->bhj', kc, b_q.view(H, hd)) * scale).reshape(R, H * Lc)
    G2 = G @ w_out                                                    # att (G W_out)^T completes x1 G^T
    b_gs = None if b_out is None else (G @ b_out)
    U = torch.einsum('bhjf,nhf->bhjn', vc, w_cout.view(d, H, hd)).reshape(R, H * Lc, d)
    G_r = G.to(wdtype)
    CS = G_r.double().sum(dim=-1).float()                             # of the ROUNDED matrix the kernels multiply with
    return {'G': G_r, 'G2': G2.to(wdtype), 'U': U.to(wdtype), 'CS': CS, 'BS': BS.contiguous(), 'b_gs': b_gs}


def folded_cross_block(t: dict, x0s: torch.Tensor, att: torch.Tensor, x1: torch.Tensor, shift: torch.Tensor, H: int,
                       eps: float = 1e-5, b_cout: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """What the device computes with the tables (host restatement for the tests): x0s [R, d] = x0 - shift (the raw fragments
    of the layer input), att [R, d] the self-attention output, x1 [R, d], shift [R].  Returns x2 [R, d]."""
    R, HL, d = t['G'].shape
    G, G2, U = t['G'].float(), t['G2'].float(), t['U'].float()
    s_raw = torch.einsum('bk,bnk->bn', x0s, G) + torch.einsum('bk,bnk->bn', att, G2)
    if t['b_gs'] is not None:
        s_raw = s_raw + t['b_gs']
    mean = x1.mean(dim=1)
    rstd = (x1.var(dim=1, unbiased=False) + eps).rsqrt()
    s = rstd[:, None] * (s_raw - (mean - shift)[:, None] * t['CS']) + t['BS']
    p = torch.softmax(s.view(R, H, HL // H), dim=-1).reshape(R, HL)
    x2 = x1 + torch.einsum('bn,bnk->bk', p, U)
    return x2 if b_cout is None else x2 + b_cout
