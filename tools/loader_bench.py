"""Ray-shard input pipeline throughput: native reader threads (csrc/r2l_shard_reader.hip) vs the obvious Python loop
(np.load x N_rand + concatenate [+ pin_memory], what a DataLoader worker does per batch).  Page-cache-hot files."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2l_amd import data  # noqa: E402


def main(n_files=400, per_batch=20, batches=100):
    d = tempfile.mkdtemp(prefix="r2l_shards_")
    rng = np.random.RandomState(0)
    rows = rng.rand(n_files * 4096, 9).astype(np.float32)
    data.write_ray_shards(rows, d, 0)
    files = data.list_ray_shards(d)
    cuda = torch.cuda.is_available()
    t0 = time.perf_counter()
    for b in range(batches):
        arrs = [np.load(files[(b * per_batch + j) % n_files]) for j in range(per_batch)]
        t = torch.from_numpy(np.concatenate(arrs, 0))
        if cuda:
            t = t.pin_memory().to("cuda", non_blocking=True)
    if cuda:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("python loop      : %7.0f files/s  %6.2f ms/batch(%d files)  %5.2f GB/s" %
          (batches * per_batch / dt, dt / batches * 1e3, per_batch, batches * per_batch * 147456 / dt / 1e9))
    for threads in (1, 2, 4, 8):
        ld = data.RayShardLoader(files, per_batch, threads=threads, device="cuda" if cuda else None)
        ld.next()
        t0 = time.perf_counter()
        for b in range(batches):
            x = ld.next()
        if cuda:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ld.close()
        print("native %d threads : %7.0f files/s  %6.2f ms/batch(%d files)  %5.2f GB/s  -> %s" %
              (threads, batches * per_batch / dt, dt / batches * 1e3, per_batch,
               batches * per_batch * 147456 / dt / 1e9, x.device))


if __name__ == "__main__":
    main()
