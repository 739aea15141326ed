#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_raymarch.sh into profiles/raymarch_pmc.json (what bench.py's roofline.traffic reads).

    python tools/pmc_raymarch_json.py <FETCH_SIZE csv> <WRITE_SIZE csv> <tag> > raymarch_pmc.json

tools/pmc_raymarch.py launches the final march 5 times through the sort permutation, then 5 times dense; the last launch of each
group is taken (warm caches do not matter: 430 MB per launch exceeds the 256 MB Infinity Cache).  FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; the counters are in KiB."""
import csv
import json
import sys


def values(path, counter):
    out = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if 'raymarch_fwd_kernel' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                out.append(float(row['Counter_Value']))
    return out


def main():
    fetch, write, tag = values(sys.argv[1], 'FETCH_SIZE'), values(sys.argv[2], 'WRITE_SIZE'), sys.argv[3]
    assert len(fetch) >= 10 and len(write) >= 10, (len(fetch), len(write))
    f_perm, f_dense, w_perm = fetch[4], fetch[9], write[4]
    R, S = 16384, 192
    alg = R * (S * 34 * 4 + 34 * 4)
    json.dump({
        'kernel': 'raymarch_fwd_kernel<3> (S=192, C=32, 16384 rays, through the sort permutation)',
        'how': 'tools/pmc_raymarch.sh: separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; FETCH_SIZE doubled as '
               'MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; counters are in KiB',
        'fetch_size_kib_raw': f_perm, 'write_size_kib_raw': w_perm,
        'hbm_bytes_per_16384_rays': (2 * f_perm + w_perm) * 1024,
        'algorithmic_bytes_per_16384_rays': alg,
        'dense_variant_fetch_kib_raw': f_dense,
        'dense_variant_fetch_bytes_doubled': 2 * f_dense * 1024,
        'collected': tag}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
