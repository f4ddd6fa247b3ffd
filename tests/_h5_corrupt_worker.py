"""Worker of tests/test_h5_corruption.py: opens seeded corruptions of one HDF5 fixture through the
product's reader (peakachu_amd.io.open_map -> cool.CoolFile -> h5lite) and reads everything the
scoring drivers read.  Prints one line per case: `<case> ok` | `<case> exc <Type>` | `<case> HANG`.
Address space is capped so that a corrupt size field ends in MemoryError, not in the OOM killer."""
import os
import resource
import signal
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peakachu_amd import io  # noqa: E402


class Hang(BaseException):
    pass


def _alarm(signum, frame):
    raise Hang()


def corrupt(blob, rng, kind):
    b = bytearray(blob)
    n = len(b)
    if kind == 0:      # single bit flips, a few of them
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # truncation
        del b[int(rng.integers(0, n)):]
    elif kind == 2:    # a block of random bytes
        at, ln = int(rng.integers(0, n)), int(rng.integers(1, 4096))
        b[at:at + ln] = rng.integers(0, 256, size=len(b[at:at + ln]), dtype=np.uint8).tobytes()
    elif kind == 3:    # a block of zeros / of 0xff (what a torn write leaves)
        at, ln = int(rng.integers(0, n)), int(rng.integers(1, 8192))
        b[at:at + ln] = bytes([0 if rng.integers(0, 2) else 255]) * len(b[at:at + ln])
    else:              # bit flips confined to the first 4 KiB (superblock, root group, first headers)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, min(n, 4096)))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def read_everything(path, uri_tail):
    lib = io.open_map(path + uri_tail)
    for chrom in lib.chromnames:
        for balance in (False, "weight"):
            m = lib.matrix(balance=balance, sparse=True).fetch(chrom)
            m.tocsr().sum()
        lib.bins().fetch(chrom)["weight"].values.sum()
    return len(lib.chromnames)


def main():
    src, uri_tail, out_dir, first, count = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    resource.setrlimit(resource.RLIMIT_AS, (6 << 30, 6 << 30))
    signal.signal(signal.SIGALRM, _alarm)
    blob = open(src, "rb").read()
    for case in range(first, first + count):
        rng = np.random.default_rng(case)
        path = os.path.join(out_dir, "case_%d%s" % (case, os.path.splitext(src)[1]))
        with open(path, "wb") as fh:
            fh.write(corrupt(blob, rng, case % 5))
        signal.setitimer(signal.ITIMER_REAL, 20.0)
        try:
            read_everything(path, uri_tail)
            res = "ok"
        except Hang:
            res = "HANG"
        except Exception as e:  # noqa: BLE001  (any Python exception is the contract)
            res = "exc %s" % type(e).__name__
        finally:
            signal.setitimer(signal.ITIMER_REAL, 0)
        os.unlink(path)
        print("%d %s" % (case, res), flush=True)
    print("DONE", flush=True)


if __name__ == "__main__":
    main()
