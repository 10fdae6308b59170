"""Merge the FETCH_SIZE and WRITE_SIZE dumps of tools/pmc_dump.py (two separate rocprofv3 --pmc passes of the same command)
into HBM bytes per launch:  bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 -- on gfx950 FETCH_SIZE reports half the bytes
of a wide coalesced stream (MI355X_MICROARCH.md, HBM section; calibrated on the BatchNorm apply kernels, whose byte counts
are exact).   python tools/pmc_traffic.py fetch.txt write.txt > profiles/rNN_pmc_traffic.txt"""
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("vtxg::", "").replace("unsigned short", "bf16")
    return re.sub(r"\(.*\)$", "", name).strip()


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"\s+%s\s+n=\s*(\d+)\s+avg=([0-9.e+]+)" % counter, line)
        if m and name:
            out[name] = (int(m.group(1)), float(m.group(2)))
        elif line and not line.startswith(" ") and not line.startswith("["):
            name = short(line)
    return out


fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
rows = []
for k, (n, f) in fetch.items():
    if k in write:
        w = write[k][1]
        rows.append(((2 * f + w) * 1024 * n, n, f, w, k))
for total, n, f, w, k in sorted(rows, reverse=True):
    if (2 * f + w) * 1024 < 5e6:
        continue
    print(f"n={n:4d}  FETCH_SIZE {f:10.1f} KiB  WRITE_SIZE {w:10.1f} KiB  -> {(2 * f + w) * 1024 / 1e6:8.1f} MB/launch   {k}")
