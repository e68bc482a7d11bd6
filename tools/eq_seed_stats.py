"""Paired statistics of the seeded training-equivalence runs (tools/train_equivalence.py with R2L_EQ_SEEDS): reads the
`=== final held-out PSNR … over seeds [...]` blocks of the given record files, pairs the default fp16 trio with the exact-fp32
family seed by seed and prints mean, sample std, standard error and a 95 % interval of the paired difference.  CPU only.
    python tools/eq_seed_stats.py profiles/r05_train_equivalence_seeds.txt profiles/r06_train_equivalence_seeds.txt"""
import re
import statistics
import sys


def blocks(path):
    """{family: {seed: psnr}} from every summary block of one record file."""
    out = {}
    seeds = None
    for line in open(path):
        m = re.match(r"=== final held-out PSNR after \d+ steps of \d+ rays over seeds \[([\d, ]+)\]", line)
        if m:
            seeds = [int(v) for v in m.group(1).split(",")]
            continue
        m = re.match(r"\s+(.+?)\s+\d+\.\d+ \+- .*per seed: ([\d. ]+)$", line)
        if m and seeds is not None:
            vals = [float(v) for v in m.group(2).split()]
            if len(vals) == len(seeds):
                out.setdefault(m.group(1).strip(), {}).update(dict(zip(seeds, vals)))
    return out


def main(paths, a="fp16 trio (default)", b="fp32 MFMA"):
    fam = {}
    for p in paths:
        for k, v in blocks(p).items():
            fam.setdefault(k, {}).update(v)
    seeds = sorted(set(fam[a]) & set(fam[b]))
    d = [fam[a][s] - fam[b][s] for s in seeds]
    n = len(d)
    sd = statistics.stdev(d)
    se = sd / n ** 0.5
    t95 = {8: 2.365, 12: 2.201, 20: 2.093, 24: 2.069, 32: 2.040}.get(n, 2.0)  # two-sided Student t, n - 1 degrees of freedom
    print("%d seeds %s" % (n, seeds))
    for k in (a, b):
        v = [fam[k][s] for s in seeds]
        print("  %-22s %.3f +- %.3f dB (sample std; %.3f .. %.3f)" % (k, statistics.mean(v), statistics.stdev(v), min(v), max(v)))
    print("  paired difference (%s) - (%s): mean %+.3f dB, sample std %.3f, standard error %.3f, 95 %% interval %+.3f .. %+.3f dB"
          % (a, b, statistics.mean(d), sd, se, statistics.mean(d) - t95 * se, statistics.mean(d) + t95 * se))
    print("  per seed: " + " ".join("%+.2f" % v for v in d))
    print("  seeds with the default ahead: %d of %d" % (sum(v > 0 for v in d), n))


if __name__ == "__main__":
    main(sys.argv[1:])
