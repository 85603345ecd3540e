"""Golden fixtures for the MultiBandDiffusion decoder option, generated from the UNMODIFIED
reference (audiocraft/models/unet.py, audiocraft/modules/diffusion_schedule.py) on CPU:

  mbd_unet.npz       DiffusionUnet.forward of a tiny U-Net (hidden 8, depth 2, growth 2, kernel 8, stride 4, 2 residual blocks
                     per layer with dilations 1 / 2, per-layer step embeddings, codec conditioning in the bottleneck), an odd
                     input length (right padding of the encoders, cropping of the decoders)
  mbd_unet_bilstm.npz  the same with the BiLSTM bottleneck
  mbd_process.npz    NoiseSchedule.generate_subsampled over a 6-step sub-sampled chain of a 100-step power schedule with a
                     MultiBandProcessor (4 bands) whose statistics are fixed, the reference's `torch.randn_like` draws
                     recorded (a device cannot share the CPU generator stream)

  mbd_model.npz      MultiBandDiffusion.generate (two bands: two U-Nets, schedules and processors; initial noise and step noise
                     draws recorded) and MultiBandDiffusion.re_eq (8 bands), run from the reference's own source file
                     (audiocraft/models/multibanddiffusion.py; its imports of the solver / loader packages, which need
                     third-party modules that are absent here and are not used by these two methods, replaced by placeholders)

`julius` is absent here: the reference's MultiBandProcessor is given oracle.mbd.split_bands in its
place, so these fixtures pin everything AROUND the band splitting (the processor's rescaling, the schedule, the U-Net) to
the reference, and the band splitting itself stays parity-unpinned (oracle/mbd.py header).

Run in the build container only:   python tests/golden/make_mbd_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the import stubs)

from oracle import mbd as ombd  # noqa: E402


class _SplitBands(torch.nn.Module):
    """stand-in for julius.SplitBands with the oracle's restatement inside"""
    def __init__(self, sample_rate, n_bands):
        super().__init__()
        self.sample_rate, self.n_bands = sample_rate, n_bands

    def forward(self, x):
        return ombd.split_bands(x, self.sample_rate, self.n_bands)


sys.modules['julius'].SplitBands = _SplitBands
from audiocraft.models.unet import DiffusionUnet  # noqa: E402
from audiocraft.modules.diffusion_schedule import MultiBandProcessor, NoiseSchedule  # noqa: E402

UNET = dict(chin=1, hidden=8, depth=2, growth=2., max_channels=10000, num_steps=100, emb_all_layers=True, bilstm=False,
            codec_dim=16, kernel=8, stride=4, norm_groups=4, res_blocks=2)


def build_unet(cfg, seed):
    torch.manual_seed(seed)
    kw = {k: v for k, v in cfg.items() if k not in ('chin', 'num_steps')}
    m = DiffusionUnet(chin=cfg['chin'], num_steps=cfg['num_steps'], **kw).eval()
    with torch.no_grad():   # GroupNorm affine parameters away from (1, 0)
        for k, p in m.named_parameters():
            if 'norm' in k:
                p.add_(0.2 * torch.randn_like(p))
    return m


def make_unet(name, cfg, seed):
    m = build_unet(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(3, 1, 203, generator=g)
    cond = torch.randn(3, cfg['codec_dim'], 9, generator=g)
    step = torch.tensor([7, 0, 99])
    with torch.no_grad():
        est = m(x, step, condition=cond).sample
        est_int = m(x[:1], 42, condition=cond[:1]).sample
    mg.save(name, cfg, m.state_dict(), x=x, condition=cond, step=step, estimate=est, estimate_step42=est_int)


def make_process():
    cfg = dict(UNET)
    m = build_unet(cfg, 11)
    sched = dict(beta_t0=1e-5, beta_t1=2.9e-2, beta_exp=7.5, num_steps=100, variance='beta', clip=5., rescale=1., noise_scale=0.9)
    proc = MultiBandProcessor(n_bands=4, sample_rate=16000, num_samples=10, power_std=0.8)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        proc.counts.fill_(37.)
        proc.sum_x.copy_(37 * 0.01 * torch.randn(4, generator=g))
        proc.sum_x2.copy_(37 * (0.05 + torch.rand(4, generator=g)))
        proc.sum_target_x2.copy_(37 * (0.1 + torch.rand(4, generator=g)))
    ns = NoiseSchedule(**sched, sample_processor=proc, device='cpu')
    initial = torch.randn(2, 1, 160, generator=g)
    cond = torch.randn(2, cfg['codec_dim'], 7, generator=g)
    step_list = [99, 80, 60, 40, 20, 0]
    noises = []
    real_randn_like = torch.randn_like

    def recorded(t, *a, **k):
        n = torch.randn(t.shape, generator=g)
        noises.append(n)
        return n
    torch.randn_like = recorded
    try:
        out = ns.generate_subsampled(m, initial=initial, step_list=step_list, condition=cond)
    finally:
        torch.randn_like = real_randn_like
    full = dict(cfg, schedule=sched, processor=dict(n_bands=4, sample_rate=16000, power_std=0.8), step_list=step_list)
    mg.save('mbd_process', full, m.state_dict(), initial=initial, condition=cond, noises=torch.stack(noises), sample=out,
            proc_counts=proc.counts, proc_sum_x=proc.sum_x, proc_sum_x2=proc.sum_x2, proc_sum_target_x2=proc.sum_target_x2,
            projected=proc.project_sample(out))


def reference_mbd_classes():
    """MultiBandDiffusion / DiffusionProcess executed from the reference's own source text.  Importing the module pulls
    audiocraft.solvers (flashy, ...) and the hub loaders; neither is touched by `generate` / `re_eq`."""
    from oracle import refstubs
    src = open(os.path.join(refstubs.REF_ROOT, 'audiocraft', 'models', 'multibanddiffusion.py')).read()
    lines = []
    for line in src.splitlines():
        if line.startswith('from .unet'):
            line = 'from audiocraft.models.unet import DiffusionUnet'
        elif line.startswith('from ..modules.diffusion_schedule'):
            line = 'from audiocraft.modules.diffusion_schedule import NoiseSchedule'
        elif line.startswith('from .encodec'):
            line = 'CompressionModel = object'
        elif line.startswith('from ..solvers.compression'):
            line = 'CompressionSolver = None'
        elif line.startswith('from .loaders'):
            line = 'load_compression_model = load_diffusion_models = None'
        lines.append(line)
    ns: dict = {}
    exec(compile('\n'.join(lines), 'reference:audiocraft/models/multibanddiffusion.py', 'exec'), ns)
    return ns['MultiBandDiffusion'], ns['DiffusionProcess']


class _Codec:
    """what MultiBandDiffusion.__init__ / generate / re_eq read from the compression model"""
    sample_rate, frame_rate, channels = 16000, 50, 1

    def parameters(self):
        return iter([torch.zeros(1)])


def make_model():
    MultiBandDiffusion, DiffusionProcess = reference_mbd_classes()
    cfg = dict(UNET, res_blocks=1)
    sched = dict(beta_t0=1e-4, beta_t1=0.1, beta_exp=1., num_steps=100, variance='beta', clip=5., rescale=1., noise_scale=1.0)
    g = torch.Generator().manual_seed(17)
    DPs, sds, pstates = [], {}, {}
    for i in range(2):
        m = build_unet(cfg, 20 + i)
        proc = MultiBandProcessor(n_bands=2, sample_rate=16000, num_samples=1, power_std=1.0)
        with torch.no_grad():
            proc.counts.fill_(5.)
            proc.sum_x.copy_(5 * 0.01 * torch.randn(2, generator=g))
            proc.sum_x2.copy_(5 * (0.5 + torch.rand(2, generator=g)))
            proc.sum_target_x2.copy_(5 * (0.5 + torch.rand(2, generator=g)))
        DPs.append(DiffusionProcess(model=m, noise_schedule=NoiseSchedule(**sched, sample_processor=proc, device='cpu')))
        sds.update({f'dp{i}.{k}': v for k, v in m.state_dict().items()})
        pstates.update({f'proc{i}_{k}': getattr(proc, k) for k in ('counts', 'sum_x', 'sum_x2', 'sum_target_x2')})
    mbd = MultiBandDiffusion(DPs=DPs, codec_model=_Codec())
    emb = torch.randn(2, cfg['codec_dim'], 6, generator=g)
    step_list = [99, 66, 33, 0]
    draws = []
    real_randn_like = torch.randn_like

    def recorded(t, *a, **k):
        draws.append(torch.randn(t.shape, generator=g))
        return draws[-1]
    torch.randn_like = recorded
    try:
        with torch.no_grad():
            wav = mbd.generate(emb, step_list=step_list)
    finally:
        torch.randn_like = real_randn_like
    ref = 0.3 * torch.randn(wav.shape, generator=g)
    with torch.no_grad():
        eq = mbd.re_eq(wav, ref, n_bands=8)
        eq_half = mbd.re_eq(wav, ref, n_bands=8, strictness=0.5)
    full = dict(cfg, schedule=sched, processor=dict(n_bands=2, sample_rate=16000, power_std=1.0), step_list=step_list,
                draws_per_band=len(draws) // 2)
    mg.save('mbd_model', full, sds, emb=emb, draws=torch.stack(draws), generated=wav, reference_wav=ref, re_eq=eq, re_eq_half=eq_half,
            **pstates)


if __name__ == '__main__':
    make_model()
    make_unet('mbd_unet', UNET, 3)
    make_unet('mbd_unet_bilstm', dict(UNET, bilstm=True, res_blocks=1), 4)
    make_process()
