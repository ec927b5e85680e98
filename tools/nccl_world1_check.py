"""Development check (GPU box, one GPU): the torch.distributed calls bench.py --gpus N makes -- gather with a gather list, all-reduce,
barrier, the pooled summaries of bayes.js_amd/shard.py -- through the "nccl" (= RCCL) backend with a world of one rank."""
import os, torch, torch.distributed as dist, sys
sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayes.js_amd")]
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
import shard
x=torch.randn(4,2,128,dtype=torch.float64,device="cuda")
gl=[torch.empty_like(x)]
shard.gather_draws(dist, x, gl, 0, equal_sizes=True)
assert torch.equal(gl[0],x)
shard.gather_draws(dist, x, gl, 0)
m,s=shard.pooled_moments(dist,x); r,e=shard.pooled_convergence(dist,x)
t=torch.tensor([1.0,2.0],dtype=torch.float64,device="cuda"); dist.all_reduce(t,op=dist.ReduceOp.MAX); dist.barrier()
print("nccl world-1 ok", m.tolist(), r.tolist())
dist.destroy_process_group()
