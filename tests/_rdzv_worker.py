"""Worker for test_launch_cpu.py (not collected): one rank of a slamhip.launch rendezvous, no GPU, no torch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))


def main() -> int:
    # only the launcher module: importing the package would not touch a GPU either, but this keeps the worker minimal
    from slamhip.launch import Rendezvous, RendezvousError, from_env

    mode = sys.argv[1]
    rank, local, world, name = from_env()
    assert name is not None and local == rank
    rz = Rendezvous(rank, world, name, timeout=float(os.environ.get("RDZV_TIMEOUT", "60")))
    try:
        got = rz.allgather({"rank": rank, "blob": bytes([rank]) * 100000})
        assert [g["rank"] for g in got] == list(range(world)) and all(g["blob"] == bytes([i]) * 100000 for i, g in enumerate(got))
        ident = rz.bcast(b"\x07" * 128 if rank == 0 else None)
        assert ident == b"\x07" * 128
        assert rz.bcast("from-last" if rank == world - 1 else None, src=world - 1) == "from-last"
        rz.barrier()
        if mode in ("spin", "spin-die"):
            # the shared-memory barrier of timing brackets: nobody passes barrier k before everybody has written k
            import time
            order = []
            for k in range(300):
                if rank == k % world:
                    time.sleep(0.0002)                 # a different straggler every time
                if mode == "spin-die" and rank == 1 and k == 100:
                    os._exit(3)
                try:
                    rz.spin_barrier()
                except RendezvousError as exc:
                    print(f"RDZV_ERROR rank {rank}: {exc}", flush=True)
                    return 5
                order.append(time.monotonic())
            stamps = rz.allgather(order)
            for k in range(299):                       # nobody leaves barrier k + 1 before everybody has entered it, i.e. left barrier k
                assert max(s_[k] for s_ in stamps) <= min(s_[k + 1] for s_ in stamps) + 1e-4, k
            t0 = time.perf_counter()
            for _ in range(500):
                rz.spin_barrier()
            per_call = (time.perf_counter() - t0) / 500
            if rank == 0:
                print(f"SPIN_US {per_call * 1e6:.1f}", flush=True)
        if mode == "skip" and rank == 1:
            return 0                                   # leaves without the collective the others are about to enter
        if mode == "die" and rank == 1:
            os._exit(3)
        try:
            worst = max(rz.allgather(float(rank)))
            assert worst == world - 1
        except RendezvousError as exc:
            print(f"RDZV_ERROR rank {rank}: {exc}", flush=True)
            return 5
    finally:
        rz.close()
    if rank == 0:
        print("RDZV_OK", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
