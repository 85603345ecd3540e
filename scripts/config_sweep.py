"""Real-time factor of the other BASELINE.json configurations (parity-test cases, not bench lines) + batch scaling
of the headline model.  Random-init weights, synthetic conditioning.  Dev / documentation tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402


def run(name, B, duration, use_sampling=True, melody=False, reps=1):
    model = MusicGen.get_random_init(name, 'cuda', torch.bfloat16)
    model.set_generation_params(use_sampling=use_sampling, top_k=250, duration=duration)
    descs = [f'prompt {i}' for i in range(B)]
    mel = torch.randn(B, 1, 32000 * 10) if melody else None

    def once():
        if melody:
            return model.generate_with_chroma(descs, mel, 32000)
        return model.generate(descs)
    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        wav = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:32s} B={B:3d} {duration:4.0f}s sampling={use_sampling!s:5s} -> {dt:6.2f} s, RTF {B * duration / dt:7.1f} '
          f'({dt / (int(duration * 50) + 3) * 1e3:.2f} ms/position)  wav {tuple(wav.shape)}', flush=True)
    del model
    torch.cuda.empty_cache()


if __name__ == '__main__':
    which = sys.argv[1:] or ['configs', 'batch']
    if 'configs' in which:
        run('facebook/musicgen-small', 1, 10, use_sampling=False)      # configs[1]
        run('facebook/musicgen-large', 8, 30)                          # configs[3], one GPU's shard
        run('facebook/musicgen-melody', 16, 30, melody=True)           # configs[4]
    if 'batch' in which:
        for B in (1, 4, 8, 16, 32):
            run('facebook/musicgen-medium', B, 10)
    if 'fold' in which:
        # the score-folded cross-attention (ACMI_CROSS_FOLD, one process per setting) where its tables are SMALL: few conditioned rows
        for name, B in (('facebook/musicgen-small', 1), ('facebook/musicgen-medium', 1), ('facebook/musicgen-medium', 2),
                        ('facebook/musicgen-medium', 4), ('facebook/musicgen-large', 1), ('facebook/musicgen-large', 4)):
            run(name, B, 10, use_sampling=name != 'facebook/musicgen-small', reps=2)
