"""Mechanics check of the multi-GPU exchange hook on ONE GPU: a 1-rank RCCL group, problem created with world_size=2/rank=0
so that the LM loop goes through every all-reduce call site (sum over one rank = identity)."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import bundler_sfm_amd as B
m, n = 60, 4000
s = B.synth_ba(m, n, 8)
calls = []
def hook(dev_ptr, count, op, _ctx):
    class _Buf:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (dev_ptr, False), "version": 2}
    t = torch.as_tensor(_Buf(), device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    calls.append((count, op))
    return 0
ref = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=B.default_options(jacobian=1, verbose=0, itmax=5))
rc0, info0 = ref.solve(); p0, _, _ = ref.download(); ref.close()
pb = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=B.default_options(jacobian=1, verbose=0, itmax=5),
               world_size=2, rank=0, nvis_global=int(s['rowptr'][-1]), nvars_global=m * 9 + 3 * n)
pb.set_allreduce(hook)
t = time.time(); rc, info = pb.solve(); t = time.time() - t
p, _, _ = pb.download()
print("hook calls", len(calls), "sizes", sorted(set(c for c, _ in calls))[:8], "time", round(t, 3))
print("info single", info0[[0, 1, 5, 6, 9]], "\ninfo hooked", info[[0, 1, 5, 6, 9]])
print("p identical:", np.array_equal(p, p0), "maxdiff", np.abs(p - p0).max())
dist.destroy_process_group()
