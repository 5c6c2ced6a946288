"""Per-rank cost of the N-GPU sharded step, rehearsed on one GPU: the staged (BN-exchange) path at the local batch
B/N of the c3_wide workload with a world-size-1 RCCL group (kernels + host staging are real; the "staged+coll" rows
also ISSUE every collective of the N-GPU step -- BatchNorm exchanges, per-block async gradient messages, likelihood
sums -- on the one-rank communicator: stream hand-offs, async handles and message order run, the wire does not)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import bench
from dun_amd.distributed import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
din, W, L, B, N_train, lr, mom, wd = bench.WORKLOADS["c3_wide"]
# usage: shard_rehearsal.py [N ...]   (default 8 4 2 1)
for N in ([int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]):
    Bl = B // N
    for sharded in (False, True, "coll", "coll-asyncfwd", "coll-localbn"):
        if N == 1 and sharded in ("coll", "coll-asyncfwd", "coll-localbn"):
            continue  # a one-GPU job has no collectives
        net = bench.build_net(din, W, L, N_train, lr, mom, wd)
        if sharded:
            # "coll-localbn": per-replica BatchNorm statistics, i.e. only the gradient messages and the likelihood sums
            spec = shard(net, sync_bn=sharded != "coll-localbn")
            spec.force_collectives = sharded in ("coll", "coll-asyncfwd", "coll-localbn")
            spec.async_forward_bn = sharded == "coll-asyncfwd"
        x, y = bench.shard_data(Bl, din, 0, torch.device("cuda"))
        for _ in range(2):
            net.fit(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            net.fit(x, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        net.fit(x, y)
        host = time.perf_counter() - t0   # time until fit returns (includes the metrics D2H sync)
        print("N=%d local batch %6d %-26s %.2f ms/step  (ideal-scaling throughput %.0f samples/s)" %
              (N, Bl, {False: "fused", True: "staged", "coll": "staged+coll", "coll-asyncfwd": "staged+coll, async fwd BN", "coll-localbn": "staged+coll, local BN"}[sharded], dt * 1e3, B / dt), flush=True)
        del net
        gc.collect()  # (module <-> buffer cycles: without it the 55 GB of the previous variant stay allocated)
        torch.cuda.empty_cache()
dist.destroy_process_group()
