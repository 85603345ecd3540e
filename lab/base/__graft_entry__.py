"""Driver entry points.

build(): compile every HIP source for gfx950 into audiocraft_amd/csrc/libacmi.so (hipcc cross-compiles
         without a GPU), import the package, and make sure the oracle (the checker) imports too.
smoke(): one tiny EnCodec encode/decode + one tiny MusicGen LM greedy generation on cuda:0 through the
         HIP kernels, each checked against the CPU oracle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    from audiocraft_amd import build as _b
    path = _b.build(force=False)
    assert os.path.exists(path)
    import audiocraft_amd  # noqa: F401
    from audiocraft_amd import _C
    assert _C.version() >= 100
    for name in _C.EXPORTS:
        assert hasattr(_C.lib, name), f"missing export {name}"
    # the oracle is pure python (torch CPU); building the checker == importing it
    import oracle.codec, oracle.lm, oracle.patterns  # noqa: F401,E401
    print(f"built {path}; acmi version {_C.version()}")


def smoke() -> None:
    import torch
    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    torch.cuda.set_device(0)
    from audiocraft_amd.models import builders
    from oracle import codec as ocodec
    from oracle import lm as olm

    # --- EnCodec (32 kHz geometry, narrow): encode -> codes -> decode vs oracle
    torch.manual_seed(0)
    ccfg = dict(builders.ENCODEC_32KHZ)
    ccfg['seanet'] = dict(ccfg['seanet'], n_filters=8)
    codec = builders.get_compression_model(ccfg, 'cuda')
    sd = {k: v.detach().cpu() for k, v in codec.state_dict().items()}
    oc = ocodec.CodecConfig(channels=1, dimension=128, n_filters=8, n_residual_layers=1, ratios=[8, 5, 4, 4],
                            causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                            sample_rate=32000, frame_rate=50)
    wav = 0.3 * torch.randn(1, 1, 6400)
    lat_ref = ocodec.seanet_encoder(sd, oc, wav)
    codes_ref = ocodec.rvq_encode(lat_ref, ocodec.codebooks_from_state(sd, 4))
    assert torch.equal(codec.quantizer.encode(lat_ref.cuda()).cpu(), codes_ref), "RVQ codes not bit-exact"
    dec = codec.decode(codes_ref.cuda()).cpu()
    dec_ref = ocodec.encodec_decode(sd, oc, codes_ref)
    err = (dec - dec_ref).abs().max().item()
    assert err < 1e-4, f"EnCodec decode max abs err {err}"

    # --- MusicGen LM (tiny), greedy, fp32 mode: tokens identical to the oracle
    lm = builders.get_lm_model(dict(dim=64, num_heads=4, num_layers=2, n_q=4, card=256, cfg_coef=3.0,
                                    conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 32,
                                                                  'length': 5}},
                                    fuser={'cross': ['description']}), 'cuda', torch.float32)
    sdl = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    cross = torch.randn(4, 5, 64, generator=g)
    cross[2:] = 0
    ct = {'description': (cross.cuda(), torch.ones(4, 5, dtype=torch.int64).cuda())}
    toks = lm.generate(None, [], num_samples=2, max_gen_len=12, use_sampling=False, condition_tensors=ct)
    ref = olm.generate(sdl, olm.LMConfig(dim=64, num_heads=4, num_layers=2, n_q=4, card=256), None, 2, cross,
                       max_gen_len=12, use_sampling=False)
    assert torch.equal(toks.cpu(), ref), "greedy tokens differ from the oracle"
    wav_out = codec.decode(toks.clamp(max=2047))
    assert wav_out.shape == (2, 1, 12 * 640)
    torch.cuda.synchronize()
    print(f"smoke ok: EnCodec decode err {err:.2e}, LM greedy tokens identical, wav {tuple(wav_out.shape)}")


if __name__ == '__main__':
    build()
    if len(sys.argv) > 1 and sys.argv[1] == 'smoke':
        smoke()
