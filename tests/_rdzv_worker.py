"""Worker of tests/test_dist.py: one rank of a rendezvous that goes wrong on purpose.
usage: _rdzv_worker.py <mode>   (rank and world size from the environment)
  ok         three operations, exit 0
  peer_dies  the LAST rank ends abruptly after the first operation
  root_dies  rank 0 ends abruptly after the first operation
Survivors must notice in their next operation (RendezvousError) and leave with code 7."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peakachu_amd.rendezvous import Rendezvous, RendezvousError  # noqa: E402

mode = sys.argv[1]
r = Rendezvous(timeout=60)
try:
    got = r.all_gather(b"r%d" % r.rank)
    assert got == [b"r%d" % q for q in range(r.world)], got
    if mode == "peer_dies" and r.rank == r.world - 1:
        os._exit(3)
    if mode == "root_dies" and r.rank == 0:
        os._exit(4)
    assert r.broadcast(b"from-root" if r.rank == 0 else None) == b"from-root"
    parts = r.gather(bytes([r.rank]) * (r.rank + 1))
    assert (parts == [bytes([q]) * (q + 1) for q in range(r.world)]) if r.rank == 0 else parts is None
    r.barrier()
except RendezvousError as e:
    sys.stderr.write("rank %d: RendezvousError: %s\n" % (r.rank, e))
    sys.exit(7)
r.close()
