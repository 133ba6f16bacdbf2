"""PCIe-inclusive rate: 2^16 x range_check(0, 2^254) emitted in chunks of 2^12 and streamed to pinned host memory."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
from plonk_gadgets_amd.host_pipeline import HostPipeline


def main():
    eng = pg.Engine(0)
    total, chunk = 1 << 16, 1 << 12
    wit = torch.from_numpy(synth.random_scalars(total).view(np.int64)).to("cuda:0")
    pipe = HostPipeline(eng, pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254), chunk)
    pipe.run(wit[:2 * chunk])
    torch.cuda.synchronize()
    t = time.perf_counter()
    pipe.run(wit)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(json.dumps({"witnesses": total, "chunk": chunk, "seconds": round(dt, 4), "gb_to_host": pipe.bytes_per_chunk() * (total // chunk) / 1e9,
                      "pcie_gbps": pipe.bytes_per_chunk() * (total // chunk) / dt / 1e9, "constraints_per_s": total * 1031 / dt}))


if __name__ == "__main__":
    main()
