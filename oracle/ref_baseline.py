"""TEST / BENCH INFRASTRUCTURE ONLY -- times the UNMODIFIED reference (`/root/reference`, imported through
oracle/refstubs.py) on the host cores for bench.py's `cpu_baseline` leg (`"kind": "reference"`).

Only usable where `/root/reference` exists (the build container); on the GPU box `available()` is False and bench.py
times the oracle port instead (`"kind": "port"`).  Nothing of the product imports this module.

What is timed is the body of the reference's autoregressive loop, exactly as `LMModel.generate` runs it
(audiocraft/models/lm.py:536-565): `LMModel._sample_next_token` inside `with lm.streaming()` -- `[seq; seq]` CFG batch,
transformer forward on the streaming state (its `torch.cat` KV cache included), CFG mix, softmax / top-k / multinomial.
"""
import os
import time
import typing as tp

import torch

REF_ROOT = os.environ.get('AUDIOCRAFT_REFERENCE', '/root/reference')


def available() -> bool:
    """True where the reference tree exists.  Importing this module never touches it: oracle/refstubs.py (which installs
    the import stubs and REQUIRES the tree) is imported by the functions below, i.e. only after `available()` said yes."""
    return os.path.isdir(os.path.join(REF_ROOT, 'audiocraft'))


def build_reference_lm(sd: tp.Optional[dict], dim: int, num_heads: int, num_layers: int, n_q: int, card: int, cross_attention: bool):
    """The reference `LMModel` with MusicGen's configuration (config/model/lm/musicgen_lm.yaml over default.yaml) and the
    given reference-format state dict (transformer / embeddings / heads; conditioner weights are not on the timed path)."""
    from . import refstubs  # noqa: F401  (installs the import stubs; needs the reference tree)
    from audiocraft.models.lm import LMModel
    from audiocraft.modules.codebooks_patterns import DelayedPatternProvider
    from audiocraft.modules.conditioners import ConditionFuser, ConditioningProvider, TextConditioner

    class _Text(TextConditioner):   # never called: the condition TENSORS are handed to _sample_next_token directly
        def tokenize(self, x):
            return x

        def forward(self, x):
            raise RuntimeError("not on the timed path")

    fuse = {'cross': ['description'] if cross_attention else [], 'prepend': [] if cross_attention else ['description'],
            'sum': [], 'input_interpolate': []}
    lm = LMModel(DelayedPatternProvider(n_q, delays=list(range(n_q))), ConditioningProvider({'description': _Text(768, dim)}),
                 ConditionFuser(fuse), n_q=n_q, card=card, dim=dim, num_heads=num_heads, hidden_scale=4, norm='layer_norm',
                 norm_first=True, bias_proj=False, weight_init=None, depthwise_init=None, zero_bias_init=False, cfg_coef=3.0,
                 num_layers=num_layers, dropout=0., activation='gelu', bias_ff=False, bias_attn=False, causal=True,
                 custom=False, memory_efficient=True, attention_as_float32=False, cross_attention=cross_attention,
                 positional_embedding='sin').eval()
    if sd is None:    # the reference's own random initialisation (calibration runs: scripts/cpu_calibration.py)
        return lm
    own = lm.state_dict()
    missing = [k for k in own if k not in sd and not k.startswith('condition_provider.')]
    assert not missing, f"state dict lacks {missing[:4]}"
    lm.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}, strict=False)
    return lm


@torch.no_grad()
def time_reference_positions(lm, B: int, cross: torch.Tensor, top_k: int, early_steps: int, late_steps: int,
                             late_context: int, generator=None) -> tp.Tuple[float, float]:
    """-> (seconds per position at the start of the stream, seconds per position at context `late_context`), CFG batch of
    2B rows, top-k sampling, each after one untimed call."""
    K, card = lm.n_q, lm.card
    H, hd = lm.transformer.layers[0].self_attn.num_heads, lm.dim // lm.transformer.layers[0].self_attn.num_heads
    g = generator or torch.Generator().manual_seed(0)
    conds = {'description': (cross, torch.ones(cross.shape[:2], dtype=torch.int64))}
    tok = torch.randint(0, card, (B, K, 1), generator=g)

    def step():
        return lm._sample_next_token(tok, conds, {}, use_sampling=True, temp=1.0, top_k=top_k, top_p=0.0, cfg_coef=3.0)

    with lm.streaming():
        step()
        t0 = time.perf_counter()
        for _ in range(early_steps):
            step()
        t_early = (time.perf_counter() - t0) / early_steps
    with lm.streaming():
        step()   # creates every streaming-state entry with the right keys / layouts; then grow the caches to late_context
        for layer in lm.transformer.layers:
            st = layer.self_attn._streaming_state
            pk = st['past_keys']
            assert pk.shape[0] == 2 * B and pk.shape[1] == H and pk.shape[3] == hd, pk.shape   # [2B, H, t, hd]: torch backend
            st['past_keys'] = torch.randn(2 * B, H, late_context, hd, generator=g)
            st['past_values'] = torch.randn(2 * B, H, late_context, hd, generator=g)
        lm.transformer._streaming_state['offsets'] = torch.full((2 * B,), late_context, dtype=torch.long)
        step()
        t0 = time.perf_counter()
        for _ in range(late_steps):
            step()
        t_late = (time.perf_counter() - t0) / late_steps
    return t_early, t_late


@torch.no_grad()
def time_reference_codec_decode(csd: dict, B: int, frames: int, generator=None) -> float:
    """Seconds for `EncodecModel.decode` of [B, 4, frames] codes at the EnCodec-32 kHz geometry (reference modules)."""
    from . import refstubs  # noqa: F401
    from audiocraft.models.encodec import EncodecModel
    from audiocraft.modules.seanet import SEANetDecoder, SEANetEncoder
    from audiocraft.quantization.vq import ResidualVectorQuantizer
    kw = dict(channels=1, dimension=128, n_filters=64, n_residual_layers=1, ratios=[8, 5, 4, 4], activation='ELU',
              activation_params={'alpha': 1.}, norm='weight_norm', norm_params={}, kernel_size=7, residual_kernel_size=3,
              last_kernel_size=7, dilation_base=2, causal=False, pad_mode='constant', true_skip=True, compress=2, lstm=2,
              disable_norm_outer_blocks=0)
    m = EncodecModel(SEANetEncoder(**kw), SEANetDecoder(**kw, trim_right_ratio=1.0),
                     ResidualVectorQuantizer(dimension=128, n_q=4, bins=2048, kmeans_init=False),
                     frame_rate=50, sample_rate=32000, channels=1).eval()
    m.load_state_dict(csd, strict=True)
    codes = torch.randint(0, 2048, (B, 4, frames), generator=generator or torch.Generator().manual_seed(0))
    m.decode(codes[:, :, :8])
    t0 = time.perf_counter()
    m.decode(codes)
    return time.perf_counter() - t0
