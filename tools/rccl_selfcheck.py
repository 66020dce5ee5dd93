import os, sys
sys.path.insert(0, '.')
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch, torch.distributed as dist
from stereomatching_amd import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
shard.barrier()
print("max", shard.max_over_ranks(1.25, torch.device("cuda", 0)))
t = torch.arange(6, dtype=torch.int32, device="cuda").reshape(1, 2, 3)
blocks = [torch.empty_like(t)]
dist.all_gather(blocks, t)
print("gather ok", blocks[0].flatten().tolist(), dist.get_backend())
dist.destroy_process_group()
