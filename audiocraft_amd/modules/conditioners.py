"""Conditioning plumbing of the generation path (host side).

Mirrors the parts of `audiocraft.modules.conditioners` that `MusicGen.generate*` touches
(SURVEY.md section 8, rows a5 / a12 / a13): `ConditioningAttributes`, `WavCondition`, null
conditions + `ClassifierFreeGuidanceDropout`, `ConditioningProvider` (tokenize / forward /
collation) and `ConditionFuser`.  The arithmetic that belongs to the path -- the conditioners'
`output_proj` Linear (reference conditioners.py:355-360, :509-515, :548-568) -- runs in the
`acmi_linear` HIP kernel.  The embedding *models* behind the conditioners (T5 encoder, Demucs,
chroma front-end) are third-party code in the reference and stay behind this boundary: a
conditioner gets their output from a pluggable `embedder` callable.
"""
import typing as tp
from collections import defaultdict
from copy import deepcopy
from dataclasses import dataclass, field
from itertools import chain

import torch
from torch import nn

from .. import _C

ConditionType = tp.Tuple[torch.Tensor, torch.Tensor]  # condition, mask


class WavCondition(tp.NamedTuple):
    wav: torch.Tensor
    length: torch.Tensor
    sample_rate: tp.List[int]
    path: tp.List[tp.Optional[str]] = []
    seek_time: tp.List[tp.Optional[float]] = []


@dataclass
class ConditioningAttributes:
    """reference conditioners.py:78-126"""
    text: tp.Dict[str, tp.Optional[str]] = field(default_factory=dict)
    wav: tp.Dict[str, WavCondition] = field(default_factory=dict)
    joint_embed: tp.Dict[str, tp.Any] = field(default_factory=dict)
    symbolic: tp.Dict[str, tp.Any] = field(default_factory=dict)

    def __getitem__(self, item):
        return getattr(self, item)

    @property
    def text_attributes(self):
        return self.text.keys()

    @property
    def wav_attributes(self):
        return self.wav.keys()

    @property
    def attributes(self):
        return {"text": self.text_attributes, "wav": self.wav_attributes,
                "joint_embed": self.joint_embed.keys(), "symbolic": self.symbolic.keys()}

    def to_flat_dict(self):
        return {**{f"text.{k}": v for k, v in self.text.items()},
                **{f"wav.{k}": v for k, v in self.wav.items()}}

    @classmethod
    def from_flat_dict(cls, x):
        out = cls()
        for k, v in x.items():
            kind, att = k.split(".")
            out[kind][att] = v
        return out


def nullify_wav(cond: WavCondition) -> WavCondition:
    """1-sample zero waveform of length 0 (reference conditioners.py:165-181)."""
    null_wav = torch.zeros_like(cond.wav[..., :1])
    n = cond.wav.shape[0]
    return WavCondition(wav=null_wav, length=torch.tensor([0] * n, device=cond.wav.device),
                        sample_rate=cond.sample_rate, path=[None] * n, seek_time=[None] * n)


def dropout_condition(sample: ConditioningAttributes, condition_type: str, condition: str) -> ConditioningAttributes:
    """In place: wav -> null wav, text -> None (reference conditioners.py:1337-1369)."""
    if condition_type not in ('text', 'wav'):
        raise ValueError(f"dropout_condition got an unexpected condition type '{condition_type}'")
    if condition not in getattr(sample, condition_type):
        raise ValueError(f"dropout_condition received an unexpected condition '{condition}'")
    if condition_type == 'wav':
        sample.wav[condition] = nullify_wav(sample.wav[condition])
    else:
        sample.text[condition] = None
    return sample


class ClassifierFreeGuidanceDropout(nn.Module):
    """All attributes dropped together with probability p (reference conditioners.py:1427-1466).
    generate() uses p=1.0 to build the unconditional half of the CFG batch."""

    def __init__(self, p: float, seed: int = 1234):
        super().__init__()
        self.p = p
        self.rng = torch.Generator()
        self.rng.manual_seed(seed)

    def forward(self, samples: tp.List[ConditioningAttributes],
                cond_types: tp.List[str] = ["wav", "text"]) -> tp.List[ConditioningAttributes]:
        if not self.training:
            return samples
        if not (torch.rand(1, generator=self.rng).item() < self.p):
            return samples
        samples = deepcopy(samples)
        for condition_type in cond_types:
            for sample in samples:
                for condition in sample.attributes[condition_type]:
                    dropout_condition(sample, condition_type, condition)
        return samples


class AttributeDropout(nn.Module):
    """Per-attribute dropout (reference conditioners.py:1380-1424); a fresh module is in training mode, which is how
    the generation path uses it (p = 1.0 / 0.0: deterministic)."""

    def __init__(self, p: tp.Dict[str, tp.Dict[str, float]], active_on_eval: bool = False, seed: int = 1234):
        super().__init__()
        self.active_on_eval = active_on_eval
        self.rng = torch.Generator()
        self.rng.manual_seed(seed)
        self.p = {condition_type: defaultdict(lambda: 0, probs) for condition_type, probs in p.items()}

    def forward(self, samples: tp.List[ConditioningAttributes]) -> tp.List[ConditioningAttributes]:
        if not self.training and not self.active_on_eval:
            return samples
        samples = deepcopy(samples)
        for condition_type, ps in self.p.items():
            for condition, p in ps.items():
                if torch.rand(1, generator=self.rng).item() < p:
                    for sample in samples:
                        dropout_condition(sample, condition_type, condition)
        return samples


def _drop_description_condition(conditions: tp.List[ConditioningAttributes]) -> tp.List[ConditioningAttributes]:
    """Text dropped, wav kept: the middle row group of double CFG (reference conditioners.py:223-236)."""
    for condition in conditions:
        assert 'description' in condition.text.keys()
        assert 'self_wav' in condition.wav.keys()
    return AttributeDropout(p={'text': {'description': 1.0}, 'wav': {'self_wav': 0.0}})(conditions)


# ------------------------------------------------------------------------------------------ conditioners

class BaseConditioner(nn.Module):
    """Holds `output_proj` (Linear dim -> output_dim, reference conditioners.py:355-360) and applies
    it with the acmi_linear kernel."""

    def __init__(self, dim: int, output_dim: int, device=None):
        super().__init__()
        self.dim = dim
        self.output_dim = output_dim
        if self.output_dim > -1:
            self.output_proj = nn.Linear(dim, output_dim, device=device)

    def project(self, embeds: torch.Tensor, mask: torch.Tensor) -> ConditionType:
        """embeds [B, L, dim] -> (output_proj(embeds) * mask, mask)  (conditioners.py:513-515, 560-568)."""
        w = self.output_proj.weight
        dev = w.device
        B, L, D = embeds.shape
        assert D == self.dim
        flat = embeds.to(device=dev, dtype=torch.float32).reshape(B * L, D).contiguous()
        out = torch.empty(B * L, self.output_dim, device=dev, dtype=torch.float32)
        tw = self.__dict__.get('_tiled')
        if tw is None or tw[0] != (w.data_ptr(), w._version):
            tw = ((w.data_ptr(), w._version), _C.TiledWeight(w.detach().float(), torch.float32))
            self.__dict__['_tiled'] = tw
        _C.linear(flat, tw[1], out, bias=self.output_proj.bias.detach().float().contiguous())
        mask = mask.to(dev)
        out = out.view(B, L, self.output_dim) * mask.unsqueeze(-1).to(out.dtype)
        return out, mask


class TextConditioner(BaseConditioner):
    pass


class WaveformConditioner(BaseConditioner):
    pass


class T5Conditioner(TextConditioner):
    """Text -> embedding sequence (reference conditioners.py:422-515).

    The T5 encoder + tokenizer are third-party (`transformers`); they are reached through
    `embedder(texts) -> (hidden [B, L, dim] f32, attention_mask [B, L])`.  With `embedder=None` a
    HuggingFace T5 is loaded lazily (needs the weights on disk; there is no network here).
    Rows whose text is None / "" get an all-zero mask like the reference (conditioners.py:492-506).
    """
    MODELS_DIMS = {"t5-small": 512, "t5-base": 768, "t5-large": 1024, "t5-3b": 1024, "t5-11b": 1024,
                   "google/flan-t5-small": 512, "google/flan-t5-base": 768, "google/flan-t5-large": 1024}

    def __init__(self, name: str, output_dim: int, finetune: bool = False, device=None, embedder=None,
                 dim: tp.Optional[int] = None, **kwargs):
        super().__init__(dim or self.MODELS_DIMS[name], output_dim, device)
        self.name = name
        self.embedder = embedder
        self.__dict__['_t5'] = None

    def _hf_embedder(self, texts):
        if self.__dict__['_t5'] is None:
            from transformers import T5EncoderModel, T5Tokenizer
            tok = T5Tokenizer.from_pretrained(self.name)
            enc = T5EncoderModel.from_pretrained(self.name).eval().to(self.output_proj.weight.device)
            self.__dict__['_t5'] = (tok, enc)
        tok, enc = self.__dict__['_t5']
        inputs = tok(texts, return_tensors='pt', padding=True).to(self.output_proj.weight.device)
        with torch.no_grad():
            return enc(**inputs).last_hidden_state.float(), inputs['attention_mask']

    def tokenize(self, x: tp.List[tp.Optional[str]]):
        return [xi if xi is not None else "" for xi in x]

    def forward(self, entries: tp.List[str]) -> ConditionType:
        embeds, mask = (self.embedder or self._hf_embedder)(entries)
        mask = mask.clone()
        empty = [i for i, e in enumerate(entries) if e == ""]
        if empty:
            mask[empty, :] = 0
        return self.project(embeds, mask)


class SyntheticTextEmbedder:
    """Deterministic stand-in for the T5 encoder: seeded N(0,1) rows, `length` tokens per prompt
    (used by bench.py / tests where no T5 weights exist)."""

    def __init__(self, dim: int = 768, length: int = 16, seed: int = 0, lengths: tp.Optional[tp.List[int]] = None):
        self.dim, self.length, self.seed, self.lengths = dim, length, seed, lengths

    def __call__(self, texts: tp.List[str]):
        g = torch.Generator().manual_seed(self.seed)
        B = len(texts)
        e = torch.randn(B, self.length, self.dim, generator=g)
        mask = torch.ones(B, self.length, dtype=torch.int64)
        if self.lengths is not None:
            for b in range(B):
                mask[b, self.lengths[b % len(self.lengths)]:] = 0
        return e, mask


class ChromaStemConditioner(WaveformConditioner):
    """Melody conditioning (reference conditioners.py:571-759): wav -> [stems: vocals + other] -> chroma frames
    [B, n_frames, n_chroma] (argmax one-hot) -> repeat / trim to the training length -> output_proj.

    The chroma front-end runs on the device (`modules/chroma.py` -> `acmi_chroma`).  Demucs (`htdemucs`, the stem
    separation in front of it) is a third-party model that is neither in the reference tree nor installable here: it
    plugs in as `stem_separator(wav [B, 1, T], sample_rate) -> wav [B, 1, T]`; without one the chroma is taken from the
    full mix.  `embedder(WavCondition) -> chroma` replaces the whole front-end (synthetic chroma for benchmarks).  As in
    `MusicGen.get_pretrained` (musicgen.py:90-92) evaluation uses `match_len_on_eval=True` and no masking."""

    def __init__(self, output_dim: int, sample_rate: int, n_chroma: int, radix2_exp: int, duration: float,
                 match_len_on_eval: bool = True, device=None, embedder=None, stem_separator=None, **kwargs):
        super().__init__(n_chroma, output_dim, device)
        from .chroma import ChromaExtractor
        self.sample_rate = sample_rate
        self.match_len_on_eval = match_len_on_eval
        self.duration = duration
        # the reference builds it with argmax from the conditioner config (chroma_stem.argmax: true, chroma2music.yaml)
        self.__dict__['chroma'] = ChromaExtractor(sample_rate=sample_rate, n_chroma=n_chroma, radix2_exp=radix2_exp,
                                                   argmax=kwargs.get('argmax', True), device=device)
        self.winhop = self.chroma.winhop
        self.chroma_len = int(duration * sample_rate) // self.winhop + 1  # == _get_chroma_len(), conditioners.py:642-646
        self.embedder = embedder
        self.stem_separator = stem_separator
        self._use_masking = not match_len_on_eval

    def _downsampling_factor(self) -> int:
        return self.winhop

    def tokenize(self, x: WavCondition) -> WavCondition:
        return x

    @torch.no_grad()
    def _compute_wav_embedding(self, wav: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """conditioners.py:678-691: stems, then chroma; a nullified wav (1 sample) goes straight to the extractor,
        which zero pads it (all-zero frames -> one-hot on class 0, like the reference's argmax)."""
        extractor = self.__dict__['chroma']
        if extractor.fbanks.device != self.output_proj.weight.device:
            extractor.to(self.output_proj.weight.device)
        if wav.shape[-1] == 1:
            return extractor(wav[:, :1])
        dev = self.output_proj.weight.device
        if self.stem_separator is not None:
            wav = self.stem_separator(wav, sample_rate)
        # what the reference's `_get_stemmed_wav` ends with (conditioners.py:675: demucs.audio.convert_audio of the merged
        # stems to the conditioner's rate, one channel): resample on the device (acmi_resample_frac), mean over channels.
        # Without a separator the full mix takes the same route (a 2-channel melody for a stereo model, any sample rate).
        from ..data_audio_utils import convert_audio
        wav = convert_audio(wav.to(dev), sample_rate, self.sample_rate, 1)
        return extractor(wav)

    def _get_wav_embedding(self, x: WavCondition) -> torch.Tensor:
        if self.embedder is not None:
            chroma = self.embedder(x).float()
        else:
            assert all(sr == x.sample_rate[0] for sr in x.sample_rate), "All sample rates in batch should be equal."
            chroma = self._compute_wav_embedding(x.wav, x.sample_rate[0])
        B, T, _ = chroma.shape
        if self.match_len_on_eval:  # conditioners.py:737-748
            if T > self.chroma_len:
                chroma = chroma[:, :self.chroma_len]
            elif T < self.chroma_len:
                n_repeat = -(-self.chroma_len // T)
                chroma = chroma.repeat(1, n_repeat, 1)[:, :self.chroma_len]
        return chroma

    def forward(self, x: WavCondition) -> ConditionType:
        chroma = self._get_wav_embedding(x)
        if self._use_masking and x.length is not None:
            lengths = (x.length / self.winhop).to(chroma.device)
            mask = (torch.arange(chroma.shape[1], device=chroma.device)[None] < lengths[:, None]).int()
        else:
            mask = torch.ones(chroma.shape[:2], dtype=torch.int64, device=chroma.device)
        return self.project(chroma, mask)


class SyntheticChromaEmbedder:
    """Seeded one-hot chroma frames; null wavs (length 0) give all-zero frames."""

    def __init__(self, n_frames: int, n_chroma: int = 12, seed: int = 3):
        self.n_frames, self.n_chroma, self.seed = n_frames, n_chroma, seed

    def __call__(self, x: WavCondition) -> torch.Tensor:
        g = torch.Generator().manual_seed(self.seed)
        B = x.wav.shape[0]
        cls = torch.randint(0, self.n_chroma, (B, self.n_frames), generator=g)
        e = torch.nn.functional.one_hot(cls, self.n_chroma).float()
        null = (x.length.cpu() == 0).view(-1, 1, 1)
        return torch.where(null, torch.zeros_like(e), e)


# ------------------------------------------------------------------------------------------ provider / fuser

class ConditioningProvider(nn.Module):
    """reference conditioners.py:1469-1616 (text + wav conditioners)."""

    def __init__(self, conditioners: tp.Dict[str, BaseConditioner], device="cpu"):
        super().__init__()
        self.device = device
        self.conditioners = nn.ModuleDict(conditioners)

    @property
    def text_conditions(self):
        return [k for k, v in self.conditioners.items() if isinstance(v, TextConditioner)]

    @property
    def wav_conditions(self):
        return [k for k, v in self.conditioners.items() if isinstance(v, WaveformConditioner)]

    @property
    def has_wav_condition(self):
        return len(self.wav_conditions) > 0

    def tokenize(self, inputs: tp.List[ConditioningAttributes]) -> tp.Dict[str, tp.Any]:
        assert all(isinstance(x, ConditioningAttributes) for x in inputs), \
            "Got unexpected types input for conditioner! should be tp.List[ConditioningAttributes]"
        text = self._collate_text(inputs)
        wavs = self._collate_wavs(inputs)
        assert set(text.keys() | wavs.keys()).issubset(set(self.conditioners.keys())), \
            f"Got an unexpected attribute! Expected {self.conditioners.keys()}, got {text.keys(), wavs.keys()}"
        return {attribute: self.conditioners[attribute].tokenize(batch)
                for attribute, batch in chain(text.items(), wavs.items())}

    def forward(self, tokenized: tp.Dict[str, tp.Any]) -> tp.Dict[str, ConditionType]:
        return {attribute: self.conditioners[attribute](inputs) for attribute, inputs in tokenized.items()}

    def _collate_text(self, samples):
        out: tp.Dict[str, tp.List[tp.Optional[str]]] = defaultdict(list)
        for sample in samples:
            for condition in self.text_conditions:
                out[condition].append(sample.text[condition])
        return out

    def _collate_wavs(self, samples) -> tp.Dict[str, WavCondition]:
        """Mono mix-down, right zero padding to the longest wav of the batch (conditioners.py:1575-1616)."""
        out: tp.Dict[str, WavCondition] = {}
        for attribute in self.wav_conditions:
            wavs, lengths, srs, paths, seeks = [], [], [], [], []
            for sample in samples:
                wav, length, sample_rate, path, seek_time = sample.wav[attribute]
                assert wav.dim() == 3 and wav.size(0) == 1, f"expected wav [1, C, T], got {tuple(wav.shape)}"
                wavs.append(wav.mean(1, keepdim=True).flatten())
                lengths.append(length)
                srs.extend(sample_rate)
                paths.extend(path)
                seeks.extend(seek_time)
            tmax = max(w.shape[0] for w in wavs)
            stacked = torch.stack([torch.nn.functional.pad(w, (0, tmax - w.shape[0])) for w in wavs])
            out[attribute] = WavCondition(stacked.unsqueeze(1), torch.cat(lengths), srs, paths, seeks)
        return out


class ConditionFuser(nn.Module):
    """How each condition enters the LM (reference conditioners.py:1672-1763).  MusicGen uses 'cross'
    (text models) or 'prepend' (melody models).  'sum' / 'input_interpolate' conditions are added to the embedded input by
    the embedding kernel (`input_ops` -> acmi_lm_state.input_add); `cross_attention_pos_emb` adds a sinusoidal embedding to
    the cross-attention source."""
    FUSING_METHODS = ["sum", "prepend", "cross", "ignore", "input_interpolate"]

    def __init__(self, fuse2cond: tp.Dict[str, tp.List[str]], cross_attention_pos_emb: bool = False,
                 cross_attention_pos_emb_scale: float = 1.0):
        super().__init__()
        assert all(k in self.FUSING_METHODS for k in fuse2cond.keys()), \
            f"Got invalid fuse method, allowed methods: {self.FUSING_METHODS}"
        self.cross_attention_pos_emb = cross_attention_pos_emb
        self.cross_attention_pos_emb_scale = cross_attention_pos_emb_scale
        self.fuse2cond = fuse2cond
        self.cond2fuse: tp.Dict[str, str] = {}
        for fuse_method, conditions in fuse2cond.items():
            for condition in conditions:
                self.cond2fuse[condition] = fuse_method

    def _check_known(self, conditions):
        assert set(conditions.keys()).issubset(set(self.cond2fuse.keys())), \
            f"given conditions contain unknown attributes for fuser, expected {self.cond2fuse.keys()}, " \
            f"got {conditions.keys()}"

    def mixed_order(self, conditions: tp.Dict[str, ConditionType]) -> bool:
        """True when a 'sum' / 'input_interpolate' condition comes AFTER a 'prepend' one in the dict order: the reference's
        loop then adds it to the prepended rows as well (`first_call_inputs`)."""
        seen_prepend = False
        for cond_type in conditions:
            op = self.cond2fuse[cond_type]
            if op == 'prepend':
                seen_prepend = True
            elif op in ('sum', 'input_interpolate') and seen_prepend:
                return True
        return False

    def first_call_inputs(self, conditions: tp.Dict[str, ConditionType], T: int):
        """The reference's loop (conditioners.py:1730-1748) for the FIRST streaming call of T token steps, replayed on a zero
        input -- every op is additive or a concatenation, so the result splits into what is prepended and what is added:
        -> (prepend [B, P, d] f32 INCLUDING what later 'sum' / 'input_interpolate' conditions add to the prepended rows,
            add [B, T, d] f32 for the call's token steps).  Later (one-step) calls see no prepend: `input_add_rows(ops, 1)`."""
        self._check_known(conditions)
        inp, P = None, 0
        for cond_type, (cond, _mask) in conditions.items():
            op = self.cond2fuse[cond_type]
            if op not in ('sum', 'input_interpolate', 'prepend'):
                continue
            cond = cond.float()
            if inp is None:
                inp = torch.zeros(cond.shape[0], T, cond.shape[2], device=cond.device)
            if op == 'prepend':
                inp = torch.cat([cond, inp], dim=1)
                P += cond.shape[1]
            else:
                inp = inp + self.input_add_rows([(op, cond)], P + T)
        if inp is None:
            return None, None
        return (inp[:, :P].contiguous() if P else None), inp[:, P:].contiguous()

    def fuse(self, conditions: tp.Dict[str, ConditionType], allow_mixed: bool = False) -> tp.Tuple[tp.Optional[torch.Tensor],
                                                                                                    tp.Optional[torch.Tensor]]:
        """-> (prepend [B, P, d] | None, cross_src [B, Lc, d] | None).

        Same ordering as the reference loop (conditioners.py:1730-1748): 'cross' conditions are
        concatenated in dict order; every 'prepend' condition is put IN FRONT of what has been
        built so far, so with the provider's dict order {description, self_wav} the prefix is
        [self_wav ; description].  'sum' / 'input_interpolate' conditions are returned by `input_ops`."""
        self._check_known(conditions)
        prepend = None
        cross = None
        for cond_type, (cond, _mask) in conditions.items():
            op = self.cond2fuse[cond_type]
            if op == 'prepend':
                prepend = cond if prepend is None else torch.cat([cond, prepend], dim=1)
            elif op == 'cross':
                cross = cond if cross is None else torch.cat([cross, cond], dim=1)
            elif op in ('sum', 'input_interpolate'):
                if prepend is not None and not allow_mixed:
                    # the reference adds this condition to the already prepended rows as well (its loop works on the
                    # concatenated input); the provider's dict order (text, then wav, then joint conditions) decides.
                    # LMModel handles it through first_call_inputs (generate, forward, streaming; not two_step_cfg)
                    raise NotImplementedError(f"'{op}' condition {cond_type!r} after a 'prepend' condition in the provider's order")
            elif op == 'ignore':
                continue
            else:
                raise ValueError(f"unknown op ({op})")
        if self.cross_attention_pos_emb and cross is not None:
            # conditioners.py:1750-1757: create_sin_embedding (transformer.py:70-89) of the source positions, default period
            L, d = cross.shape[1], cross.shape[2]
            half = d // 2
            pos = torch.arange(L, device=cross.device, dtype=torch.float32).view(1, -1, 1)
            adim = torch.arange(half, device=cross.device, dtype=torch.float32).view(1, 1, -1)
            phase = pos / (torch.full([], 10000., device=cross.device) ** (adim / (half - 1)))
            cross = cross + self.cross_attention_pos_emb_scale * torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1).to(cross)
        return prepend, cross

    def input_ops(self, conditions: tp.Dict[str, ConditionType]) -> tp.List[tp.Tuple[str, torch.Tensor]]:
        """The 'sum' / 'input_interpolate' conditions in dict order: [(op, cond [B, Tc, d])] (conditioners.py:1733-1737)."""
        self._check_known(conditions)
        return [(self.cond2fuse[k], cond) for k, (cond, _mask) in conditions.items()
                if self.cond2fuse[k] in ('sum', 'input_interpolate')]

    @staticmethod
    def input_add_rows(ops: tp.Sequence[tp.Tuple[str, torch.Tensor]], T: int) -> tp.Optional[torch.Tensor]:
        """What the reference's fuser adds to the embedded input of ONE call of T steps: [B, T, d] f32 (None without ops).
        'sum': the condition itself, one frame broadcast over the call or one frame per step (its in-place `input += cond`
        allows nothing else); 'input_interpolate': F.interpolate(cond, size=T), i.e. frame min(floor(t * f32(Tc / T)), Tc - 1)
        for step t -- a gather, evaluated in f32 like ATen's nearest kernel."""
        total = None
        for op, cond in ops:
            cond = cond.float()
            Tc = cond.shape[1]
            if op == 'sum':
                if Tc not in (1, T):
                    raise RuntimeError(f"'sum' condition of {Tc} frames cannot be added to a call of {T} steps "
                                       "(the reference's in-place add fails the same way)")
                add = cond.expand(-1, T, -1)
            else:
                scale = torch.tensor(Tc, dtype=torch.float32) / T
                idx = torch.floor(torch.arange(T, dtype=torch.float32) * scale).long().clamp_max(Tc - 1)
                add = cond[:, idx.to(cond.device)]
            total = add if total is None else total + add
        return total
