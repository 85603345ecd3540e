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
