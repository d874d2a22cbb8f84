"""Probe: does this RCCL build accept two ranks on ONE GPU (so that the multi-rank paths could be exercised on the one-GPU
box)?  Run under torch.distributed.run --nproc-per-node 2; prints what happened and exits."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', rank=rank, world_size=2, device_id=torch.device('cuda', 0),
                            timeout=datetime.timedelta(seconds=40))
    t = torch.ones(4, device='cuda') * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f'rank {rank}: all_reduce ok {t.tolist()}', flush=True)
    buf = torch.full((1 << 20,), float(rank), dtype=torch.float64, device='cuda')
    inbox = torch.empty_like(buf)
    ops = [dist.P2POp(dist.isend, buf, 1 - rank), dist.P2POp(dist.irecv, inbox, 1 - rank)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print(f'rank {rank}: p2p ok {inbox[0].item()}', flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f'rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}', flush=True)
    sys.exit(0)
