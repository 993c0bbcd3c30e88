"""Write the flat weight blob (weights + derived host tables) for a BASELINE config to a file, for
non-Python hosts of the C ABI (tools/ymt3_run.cpp).  `python -m yourmt3_amd.export_blob out.bin [config_index]`"""
import sys

from .config import baseline_config
from .tables import derived_tables
from .weights import make_weights, pack_blob


def main():
    path = sys.argv[1]
    cfg = baseline_config(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    W = make_weights(cfg, seed=1234)
    blob = pack_blob({**W, **derived_tables(W, cfg)})
    with open(path, "wb") as f:
        f.write(blob)
    print(path, len(blob), "bytes")


if __name__ == "__main__":
    main()
