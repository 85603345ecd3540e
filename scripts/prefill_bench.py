"""What the prompt / prefix positions cost (dev / documentation tool; numbers go to profiles/ and DESIGN.md).

Two cases of the reference's path where the first streaming forward covers many positions at once
(audiocraft/models/lm.py:540-543, models/genmodel.py:233-262):

  window   MusicGen-medium, 8 prompts (16 CFG rows), a 600-token prompt: every > 30 s window of `generate` is prompted with
           the last 600 tokens of the previous one (extend_stride 18 s)
  melody   MusicGen-melody, 16 prompts (32 CFG rows), the 251-row prefix (235 chroma frames + 16 text rows)

Prints the time of the prefill alone (HIP events around LMModel._prefill) and the rows / flops it covers.
    python scripts/prefill_bench.py [window] [melody] [--reps N]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402


def timed_prefill(lm):
    """Wrap lm._prefill so that every call records its duration (ms) in lm._prefill_ms."""
    lm._prefill_ms = []
    inner = lm._prefill

    def wrapper(desc, state, n_positions, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = inner(desc, state, n_positions, *a, **kw)
        e1.record()
        e1.synchronize()
        lm._prefill_ms.append((n_positions, e0.elapsed_time(e1)))
        return out
    lm._prefill = wrapper


def flops(lm, rows):
    per_layer = (14 if lm.has_cross_attention else 12) * lm.dim * lm.dim
    return 2.0 * per_layer * lm.num_layers * rows


def run(kind, reps):
    if kind == 'window':
        name, B, T0 = 'facebook/musicgen-medium', 8, 600
    else:
        name, B, T0 = 'facebook/musicgen-melody', 16, 0
    model = MusicGen.get_random_init(name, 'cuda', torch.bfloat16)
    lm = model.lm
    timed_prefill(lm)
    model.set_generation_params(use_sampling=True, top_k=250, duration=(T0 + 8) / 50.0 if T0 else 0.2)
    descs = [f'prompt {i}' for i in range(B)]
    for _ in range(reps + 1):
        if kind == 'window':
            prompt = torch.randint(0, 2048, (B, 4, T0), device='cuda')
            attributes, _ = model._prepare_tokens_and_attributes(descs, None)
            lm.generate(prompt, attributes, max_gen_len=T0 + 8, **{k: v for k, v in model.generation_params.items()
                                                                  if k not in ('cfg_coef_beta',)})
        else:
            mel = torch.randn(B, 1, 32000 * 10)
            model.generate_with_chroma(descs, mel, 32000)
    torch.cuda.synchronize()
    calls = lm._prefill_ms[1:]                      # first generate = warm-up (graph capture, allocations)
    npos = calls[0][0]
    ms = sorted(c[1] for c in calls)[len(calls) // 2]
    rows = npos * 2 * B
    out = dict(case=kind, model=name, batch=B, cfg_rows=2 * B, positions=npos, rows=rows, prefill_ms=round(ms, 3),
               tflop=round(flops(lm, rows) / 1e12, 2), tflops_per_s=round(flops(lm, rows) / 1e9 / ms, 1),
               path=os.environ.get('ACMI_PREFILL', 'default'))
    print(json.dumps(out), flush=True)
    del model
    torch.cuda.empty_cache()


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 3
    for kind in (args or ['window', 'melody']):
        run(kind, reps)
