#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) into per-kernel HBM bytes
per launch.  Usage: summarize_pmc.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json>

Units and corrections as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so fetch bytes = 2 * FETCH_SIZE * 1024 (an upper bound for
narrow access patterns); WRITE_SIZE is taken as is."""
import csv, glob, json, os, re, sys
from collections import defaultdict

KERNELS = ["k_linearize", "k_reduce", "k_dense", "k_backsub", "k_lk_track_levels", "k_lk_track_units", "k_lk_track", "k_scharr", "k_pyr_down", "k_clahe_apply", "k_remap"]


def per_kernel(directory, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                short = next((k for k in KERNELS if re.search(r"\b%s\b" % k, name)), None)
                if short is None:
                    continue
                acc[short][0] += float(row["Counter_Value"])
                acc[short][1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items() if n}


def kernel_stats(directory):
    """per-kernel {avg_us, launches} of a `rocprofv3 --kernel-trace --stats --output-format csv` pass (its *kernel_stats.csv)"""
    st = {}
    for path in glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                short = next((k for k in KERNELS if re.search(r"\b%s\b" % k, row.get("Name", ""))), None)
                if short:
                    n0, t0 = st.get(short, (0, 0.0))
                    st[short] = (n0 + int(row["Calls"]), t0 + float(row["TotalDurationNs"]))
    return {k: {"avg_us": t / n / 1e3, "launches": n} for k, (n, t) in st.items() if n}


def source_sha256():
    """Fingerprint of the kernel sources the counters were collected on (bench.py prints `traffic` only when it matches the
    sources it runs: the GPU box has no .git, so the commit hash is not available there)"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pvio_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in KERNELS:
        if k not in fetch and k not in write:
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        kernels[k] = {
            "FETCH_SIZE_KiB_per_launch": f, "launches_FETCH_SIZE": nf,
            "WRITE_SIZE_KiB_per_launch": w, "launches_WRITE_SIZE": nw,
            "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
        }
    note = ("rocprofv3 --kernel-trace --pmc <counter> on `python bench.py --steps 20 --warmup 2 --no-cpu-baseline` (separate passes for "
            "FETCH_SIZE and WRITE_SIZE, ROCm 7.2, gfx950). Counter unit = KiB. Per MI355X_MICROARCH.md FETCH_SIZE reports 1/2 of the bytes "
            "of wide coalesced reads on gfx950 -> fetch_bytes_corrected = 2 * FETCH_SIZE * 1024 (upper bound for narrow access patterns); "
            "WRITE_SIZE is uncalibrated.")
    json.dump({"note": note, "source_sha256": source_sha256(), "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
