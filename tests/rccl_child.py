"""Child process of tests/test_gpu_rccl.py: a ONE-rank RCCL communicator (backend "nccl" on ROCm) on cuda:0 running the
collectives this package uses on multi-GPU jobs -- sharding.gather_records (gather and all-gather form, packed int32 block
carrying float32 records + int64 frame indices), sharding.broadcast_from_root of a complex64 filter, and bench.py's
gather / max-reduction / barrier -- on device tensors.  Prints one JSON line.  Started as a fresh interpreter: nothing here
runs before this process has initialised the GPU itself."""
import importlib
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    SH = importlib.import_module("channel-estimation-in-5g-network_amd.sharding")
    E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
    out = {}
    # per-frame records of a real (small) batch, gathered through the collective
    geom = E.Geometry(useful_subcarriers=72)
    eng = E.Engine(geom, 0)
    B = 37
    slots = eng.build_patterns(0, [99] * B, seed=1, pattern_ids=np.arange(B, dtype=np.uint64), method="linear")
    m = eng.run_frames(E.make_descs("EVA", 70.0, 10.0, slots, n=B), seed=2, frame_offset=(1 << 24) + 5, outputs=("metrics",))["metrics"]
    local = m.cpu().numpy()
    ids = (1 << 24) + 5 + np.arange(B, dtype=np.int64)
    rec, gid = SH.gather_records(local, ids, dst=0, capacity=B + 3, force_collective=True)
    out["gather_equal"] = bool(np.array_equal(rec, local) and np.array_equal(gid, ids) and gid.dtype == np.int64)
    rec2, gid2 = SH.gather_records(local[:, :3], ids, all_ranks=True, force_collective=True)      # capacity agreed by the extra all-gather
    out["all_gather_equal"] = bool(np.array_equal(rec2, local[:, :3]) and np.array_equal(gid2, ids))
    # config 5's broadcast of W (complex64 travels as its real view)
    W = (torch.randn(173, 173, device="cuda") + 1j * torch.randn(173, 173, device="cuda")).to(torch.complex64)
    W0 = W.clone()
    SH.broadcast_from_root(W, 0, force_collective=True)
    out["broadcast_equal"] = bool(torch.equal(W, W0))
    # bench.py's end-of-job collectives
    summ = torch.tensor([1.5, 2.5, 3.5, 4.0], dtype=torch.float64, device="cuda")
    got = [torch.zeros_like(summ)]
    dist.gather(summ, got, dst=0)
    t = torch.tensor([0.123], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    out["bench_gather_equal"] = bool(torch.equal(got[0], summ)) and abs(float(t.item()) - 0.123) < 1e-12
    out["backend"] = dist.get_backend()
    out["rccl_loaded"] = any("librccl" in ln for ln in open("/proc/self/maps"))
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
