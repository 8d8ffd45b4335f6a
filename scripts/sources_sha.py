"""sha256 (first 16 hex digits) over the library's sources — locityper_amd/csrc/*.{hip,hpp,cpp} and include/*.h, in name order. Written
next to every profile pass (scripts/gpu_round.sh) and recomputed by bench.py, so that a counter file taken from other kernels than the
ones being timed shows in the bench line (`traffic_is_current`)."""
import glob, hashlib, os, sys


def sources_sha16(root):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "locityper_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "locityper_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(root, "locityper_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(sources_sha16(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
