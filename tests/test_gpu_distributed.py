"""-m gpu, needs >= 2 MI355X: the sharded generation path on RCCL (torch.distributed 'nccl' backend over xGMI) against
the single-device result, with the REAL kernels (tests/test_distributed_cpu.py checks the same property on gloo with a
stand-in LM).  Skips on a 1-GPU box."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B_global, T, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from audiocraft_amd import distributed as adist
    from audiocraft_amd.models.musicgen import MusicGen
    adist.init_from_env('nccl')
    dev = torch.device('cuda', rank)
    torch.manual_seed(0)          # identical replicas on every rank (weights are replicated, prompts sharded)
    model = MusicGen.get_pretrained('debug', dev)
    descriptions = [f'prompt {i}' * (1 + i % 3) for i in range(B_global)]
    tokens, wav = adist.generate_sharded(model, descriptions if rank == 0 else None, B_global, T, decode=True,
                                         gather_audio=True, generation_params={'use_sampling': False})
    out[rank] = (tokens.cpu(), wav.cpu())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('B_global', [4, 5])
def test_sharded_generation_on_rccl_matches_single_device(B_global):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X")
    import torch.multiprocessing as mp
    from audiocraft_amd import distributed as adist
    from audiocraft_amd.models.musicgen import MusicGen
    world, T = 2, 20
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B_global, T, out), nprocs=world, join=True)
    torch.manual_seed(0)
    model = MusicGen.get_pretrained('debug', 'cuda')
    descriptions = [f'prompt {i}' * (1 + i % 3) for i in range(B_global)]
    ref_t, ref_w = adist.generate_sharded(model, descriptions, B_global, T, decode=True,
                                          generation_params={'use_sampling': False})
    for rank in range(world):
        tokens, wav = out[rank]
        assert torch.equal(tokens, ref_t.cpu()), f"rank {rank}: gathered tokens differ from the single-device result"
        assert torch.allclose(wav, ref_w.cpu(), atol=1e-5)


# ----------------------------------------------------------------------------------------------------------------------
# The RCCL path above needs two GPUs and the 1-GPU test box skips it.  What CAN be executed on one device is everything
# except the transport: `generate_sharded` itself, with the REAL kernels, driven through a fake 2-rank process group --
# two threads of this process, one model replica each, `torch.distributed`'s entry points replaced by an in-memory
# mailbox that moves exactly the tensors the collectives would move (the packed header + payload of the broadcast, the
# padded shards of the all-gather).  GPU work is serialised by the mailbox itself: rank 1 is released from the
# broadcast only when rank 0 has reached its all-gather.
# ----------------------------------------------------------------------------------------------------------------------
class _LoopbackGroup:
    def __init__(self, world):
        import threading
        self.world = world
        self.tls = threading.local()
        self.cv = threading.Condition()
        self.bcast = []                 # broadcast payloads in call order (src deposits, others copy)
        self.gather = {}                # all_gather call index -> {rank: tensor}
        self.rank0_in_gather = False
        self.log = []

    # -- the torch.distributed surface audiocraft_amd.distributed uses
    def is_initialized(self):
        return True

    def get_rank(self):
        return self.tls.rank

    def get_world_size(self):
        return self.world

    def broadcast(self, tensor, src=0):
        r = self.tls.rank
        idx = self.tls.n_bcast = getattr(self.tls, 'n_bcast', -1) + 1
        with self.cv:
            if r == src:
                self.bcast.append(tensor.detach().clone())
                self.log.append(('broadcast', idx, tuple(tensor.shape), str(tensor.dtype)))
                self.cv.notify_all()
            else:
                ok = self.cv.wait_for(lambda: len(self.bcast) > idx and self.rank0_in_gather, timeout=600)
                assert ok, "loopback broadcast timed out"
                assert self.bcast[idx].shape == tensor.shape and self.bcast[idx].dtype == tensor.dtype, \
                    "receiver's buffer does not match what the source sent (header / layout mismatch)"
                tensor.copy_(self.bcast[idx])

    def all_gather(self, bufs, tensor):
        r = self.tls.rank
        idx = self.tls.n_gather = getattr(self.tls, 'n_gather', -1) + 1
        with self.cv:
            self.gather.setdefault(idx, {})[r] = tensor.detach().clone()
            if r == 0:
                self.rank0_in_gather = True
            self.cv.notify_all()
            ok = self.cv.wait_for(lambda: len(self.gather[idx]) == self.world, timeout=600)
            assert ok, "loopback all_gather timed out"
            for j in range(self.world):
                assert bufs[j].shape == self.gather[idx][j].shape, "all_gather needs equal shapes on every rank"
                bufs[j].copy_(self.gather[idx][j])


def test_sharded_path_fake_two_ranks_real_kernels(monkeypatch):
    """generate_sharded on a fake 2-rank group, real kernels, ONE device: shard 0 and shard 1 of an uneven global batch
    (5 prompts -> 3 + 2), conditions through pack -> broadcast -> unpack, tokens and audio through the padded all-gather;
    the gathered result must equal the unsharded greedy result bit for bit (tokens) / to 1e-5 (audio)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import threading
    import torch.distributed as dist
    from audiocraft_amd import distributed as adist
    from audiocraft_amd.models.musicgen import MusicGen
    B_global, T, world = 5, 20, 2
    descriptions = [f'prompt {i}' * (1 + i % 3) for i in range(B_global)]
    torch.manual_seed(0)
    ref_model = MusicGen.get_pretrained('debug', 'cuda')
    ref_t, ref_w = adist.generate_sharded(ref_model, descriptions, B_global, T, decode=True,
                                          generation_params={'use_sampling': False})
    group = _LoopbackGroup(world)
    for name in ('is_initialized', 'get_rank', 'get_world_size', 'broadcast', 'all_gather'):
        monkeypatch.setattr(dist, name, getattr(group, name))
    models = []
    for _ in range(world):           # identical replicas (weights replicated, prompts sharded)
        torch.manual_seed(0)
        models.append(MusicGen.get_pretrained('debug', 'cuda'))
    out, errors = {}, []

    def run(rk):
        try:
            group.tls.rank = rk
            torch.cuda.set_device(0)
            tokens, wav = adist.generate_sharded(models[rk], descriptions if rk == 0 else None, B_global, T, decode=True,
                                                 gather_audio=True, generation_params={'use_sampling': False})
            torch.cuda.synchronize()
            out[rk] = (tokens.cpu(), wav.cpu())
        except BaseException as e:   # noqa: BLE001 -- re-raised in the main thread
            errors.append((rk, e))
            with group.cv:           # never leave the other rank waiting
                group.rank0_in_gather = True
                group.cv.notify_all()

    threads = [threading.Thread(target=run, args=(rk,)) for rk in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(900)
    assert not errors, errors
    # two broadcasts whatever the number of conditions (fixed-layout header, one payload), two all-gathers (tokens, audio)
    assert [e[0] for e in group.log] == ['broadcast', 'broadcast'] and group.log[0][3] == 'torch.int64'
    assert sorted(group.gather) == [0, 1] and all(len(v) == world for v in group.gather.values())
    assert group.gather[0][0].shape[0] == 3           # ceil(5 / 2) padded rows per rank
    for rk in range(world):
        tokens, wav = out[rk]
        assert tokens.shape == (B_global, 4, T)
        assert torch.equal(tokens, ref_t.cpu()), f"rank {rk}: gathered tokens differ from the unsharded result"
        assert torch.allclose(wav, ref_w.cpu(), atol=1e-5)


# ----------------------------------------------------------------------------------------------------------------------
# The whole multi-PROCESS path on a 1-GPU box: `python bench.py --gpus 2` exactly as the driver's scaling run starts it
# (no launcher -> bench.py spawns its ranks, rendezvous on 127.0.0.1), with the two test switches of
# audiocraft_amd/distributed.py -- ACMI_DIST_BACKEND=gloo (RCCL refuses two ranks on one device) and
# ACMI_ALLOW_SHARED_DEVICE=1 (rank r -> device r % device_count).  Executes _spawn_ranks, init_from_env, the two
# broadcasts of the conditioning, the sharded generate + decode with the real kernels in two processes, gather_rows, the
# all_reduce(MAX) of the clock and rank 0's JSON line.
# ----------------------------------------------------------------------------------------------------------------------
def _run_bench(tmp_path, gpus, batch, tag, model='facebook/musicgen-small', greedy=True, duration='1'):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tok = tmp_path / f'tokens_{tag}.pt'
    env = dict(os.environ, ACMI_DIST_BACKEND='gloo', ACMI_ALLOW_SHARED_DEVICE='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus), '--steps', '1', '--warmup', '0', '--duration', duration,
           '--model', model, '--batch', str(batch), '--no-cpu-baseline', '--no-roofline', '--dump-tokens', str(tok)]
    if greedy:
        cmd.append('--greedy')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), torch.load(tok)


def test_bench_two_processes_on_one_device(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    line2, tok2 = _run_bench(tmp_path, 2, 2, 'w2')     # 2 ranks x 2 prompts
    assert line2['n_gpus'] == 2 and line2['config']['global_batch'] == 4 and line2['scaling'] == 'weak'
    assert line2['config']['parallelism'].startswith('dp2') and line2['value'] > 0 and line2['ms_per_step'] > 0
    assert line2['step_roofline']['peak'] == 2 * 8000.0
    line1, tok1 = _run_bench(tmp_path, 1, 4, 'w1')     # the same 4 prompts on one rank
    assert line1['n_gpus'] == 1 and line1['config']['global_batch'] == 4
    assert tok2.shape == tok1.shape == (4, 4, 50)
    assert torch.equal(tok2, tok1), "gathered greedy tokens of the 2-rank run differ from the unsharded run"


def test_bench_configs3_flags_two_processes_sampled(tmp_path):
    """`bench.py --gpus N --model facebook/musicgen-large --batch 8` is how the driver's scaling run starts BASELINE.json
    configs[3]; here with 2 ranks on one device, 2 prompts per rank and 0.4 s, SAMPLED (the mode of that configuration).
    Checks the line, and the per-rank sampling seeds of generate_sharded (base_seed + rank, cf. reference utils/utils.py:203-223):
    rank 0's shard reproduces the first rows of the unsharded run (same seed, same per-sample counters: a fixed (seed, rank)
    reproduces), rank 1's shard does not reproduce the unsharded run's rows 2-3 (another seed)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    line2, tok2 = _run_bench(tmp_path, 2, 2, 'l2', model='facebook/musicgen-large', greedy=False, duration='0.4')
    assert line2['n_gpus'] == 2 and line2['config']['global_batch'] == 4 and line2['scaling'] == 'weak' and line2['dtype'] == 'bf16'
    assert 'musicgen-large' in line2['config']['workload'] and 'top-k 250' in line2['config']['workload']
    assert 'not a BASELINE.json configuration' in line2['config']['workload']    # shortened: the tag says so
    line1, tok1 = _run_bench(tmp_path, 1, 4, 'l1', model='facebook/musicgen-large', greedy=False, duration='0.4')
    assert tok2.shape == tok1.shape == (4, 4, 20)
    assert torch.equal(tok2[:2], tok1[:2]), "rank 0's shard (seed, rank fixed) must reproduce the unsharded rows"
    assert not torch.equal(tok2[2:], tok1[2:]), "rank 1 samples with its own seed"


# ----------------------------------------------------------------------------------------------------------------------
# RCCL itself on the 1-GPU box: a process group of ONE rank on the 'nccl' backend.  It cannot show scaling, but it executes
# every torch.distributed call of the sharded path against the real RCCL library with the real tensors -- the group
# creation with `device_id`, the int64 header and f32 payload broadcasts, the int64 token and f32 waveform all-gathers, the
# barrier and the float64 MAX all-reduce of bench.py's clock -- so a dtype / device / argument RCCL rejects fails HERE and
# not on the first 8-GPU run.  (gloo accepts host tensors and dtypes RCCL does not; the CPU tests cannot see that.)
# ----------------------------------------------------------------------------------------------------------------------
_RCCL_ONE_RANK = r'''
import os, sys, time, torch
import torch.distributed as dist
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1])
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
from audiocraft_amd import distributed as adist
from audiocraft_amd.models.musicgen import MusicGen
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))   # as init_from_env does for N > 1
assert dist.get_backend() == 'nccl' and adist.world_size() == 1
print('RCCL_GROUP_UP', flush=True)
calls = []
for name in ('broadcast', 'all_gather', 'barrier', 'all_reduce'):
    def wrap(fn, name=name):
        def f(*a, **k):
            t = a[1] if name == 'all_gather' else (a[0] if a else None)
            calls.append((name, None if t is None else (str(t.dtype), t.device.type, tuple(t.shape))))
            return fn(*a, **k)
        return f
    setattr(dist, name, wrap(getattr(dist, name)))
torch.manual_seed(0)
model = MusicGen.get_pretrained('debug', 'cuda')
B, T = 3, 20
desc = [f'prompt {i}' * (1 + i % 3) for i in range(B)]
tok, wav = adist.generate_sharded(model, desc, B, T, decode=True, gather_audio=True, generation_params={'use_sampling': False})
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.25], device='cuda', dtype=torch.float64)            # bench.py's max-over-ranks clock
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
dist.barrier()
dist.destroy_process_group()
assert not dist.is_initialized()
tok0, wav0 = adist.generate_sharded(model, desc, B, T, decode=True, gather_audio=True, generation_params={'use_sampling': False})
assert torch.equal(tok, tok0) and torch.equal(wav, wav0), "through RCCL != without a group"
kinds = [c[0] for c in calls]
assert kinds.count('broadcast') == 2 and kinds.count('all_gather') == 2, kinds
assert all(c[1] is None or c[1][1] == 'cuda' for c in calls), calls       # RCCL only ever saw device tensors
print('RCCL_ONE_RANK_OK', calls, flush=True)
'''


def test_rccl_single_rank_group_runs_every_collective_of_the_sharded_path(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'rccl_one_rank.py'
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'ACMI_DIST_BACKEND', 'ACMI_ALLOW_SHARED_DEVICE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), str(_free_port())], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    if 'RCCL_GROUP_UP' not in r.stdout:      # the box could not create the communicator at all: environment, not this package
        pytest.skip("RCCL could not create a one-rank group here: " + r.stderr[-400:])
    assert r.returncode == 0 and 'RCCL_ONE_RANK_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    print(r.stdout.strip().splitlines()[-1])
