"""HBM bytes per launch from the two PMC passes of scripts/profile_bench.sh:  2 * FETCH_SIZE + WRITE_SIZE  (both in KB).
FETCH_SIZE is doubled because on gfx950 it tallies 64 bytes per 128-byte request of a wide coalesced read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Kernels are matched to the names bench.py uses; launches of one
C++ kernel that move different amounts (first sweep after a spread predictor, 1-field vs M-field passes) are told apart
by their own per-dispatch counter values."""
import collections
import csv
import glob
import json
import re
import sys

out, n = sys.argv[1], int(sys.argv[2])


def per_dispatch(kind):
    rows = collections.defaultdict(dict)          # kernel -> dispatch id -> value (KB)
    for f in glob.glob(f'{out}/{kind}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '?')
            d = int(row.get('Dispatch_Id', 0))
            rows[k][d] = rows[k].get(d, 0.0) + float(row.get('Counter_Value', 0) or 0)
    return rows


fetch, write = per_dispatch('fetch'), per_dispatch('write')
summary = {kind: {k: {'sum': sum(v.values()), 'dispatches': len(v), 'per_dispatch': sum(v.values()) / max(len(v), 1)}
                  for k, v in rows.items()} for kind, rows in (('fetch', fetch), ('write', write))}
json.dump(summary, open(f'{out}/pmc_summary.json', 'w'), indent=1)

NAMES = [(r'k_spec_z<\d+, \d+, 5, \d+>', 'spec_z_res_tab'), (r'k_spec_z<\d+, \d+, [34], \d+>', 'spec_z_res_v0'), (r'k_spec_store', 'spec_store'),
         (r'k_spec_z<\d+, \d+, 1, \d+>', 'spec_z_res'), (r'k_spec_z<\d+, \d+, 0, \d+>', 'spec_z'),
         (r'k_ffty<\d+, \d+, 1>', 'fft_y_inv'), (r'k_ffty<\d+, \d+, -1>', 'fft_y_fwd'),
         (r'k_fftx_inv<\d+, \d+, true, false, false, 2(, false)?>', 'fft_x_norm2'), (r'k_fftx_inv<\d+, \d+, false, false, false, 1(, false)?>', 'fft_x_scr'),
         (r'k_fftx_inv<\d+, \d+, true, false(, false)?(, 0)?(, false)?>', 'fft_x_norm'), (r'k_fftx_norm_half', 'fft_x_norm'),
         (r'k_fftx_inv<\d+, \d+, true, false, true(, 0)?(, false)?>', 'fft_x_norm_add'),
         (r'k_fftx_inv<\d+, \d+, false, true(, false)?(, 0)?(, false)?>', 'fft_x_inv'), (r'k_fftx_inv<\d+, \d+, true, true(, false)?(, 0)?(, false)?>', 'fft_x_inv_norm'),
         (r'k_trail_z<', 'trail_z'), (r'k_trail_store<', 'trail_store'), (r'k_trail_nyq<', 'trail_nyq'),
         (r'k_fftx_fwd', 'fft_x_fwd'), (r'k_fftz_plain<\d+, 1, true>', 'fft_z_sym'), (r'k_fftz_plain<\d+, 1(, false)?>', 'fft_z_inv'),
         (r'k_fftz_plain<\d+, -1(, false)?>', 'fft_z_fwd'),
         (r'k_stencil3d<4>', 'stencil_max'), (r'k_stencil3d_res', 'stencil_res'), (r'k_spec_point', 'spec_point')]
traffic = {}
for kname in set(fetch) | set(write):
    name = next((nm for pat, nm in NAMES if re.search(pat, kname)), None)
    if name is None:
        continue
    # launches in order of dispatch within each pass (the two passes run the same command: same sequence)
    fv = [v for _, v in sorted(fetch.get(kname, {}).items())]
    wv = [v for _, v in sorted(write.get(kname, {}).items())]
    m = min(len(fv), len(wv)) if fv and wv else max(len(fv), len(wv))
    per = [(2.0 * (fv[i] if i < len(fv) else 0.0) + (wv[i] if i < len(wv) else 0.0)) * 1024.0 for i in range(m)]
    if not per:
        continue
    big = max(per)
    groups = {'': [p for p in per if p > 0.8 * big], '_small': [p for p in per if p <= 0.8 * big]}
    for suffix, vals in groups.items():
        if not vals:
            continue
        key = name + ('' if suffix == '' else ('_spread' if name.startswith('spec_z') else '_1field'))
        traffic[f'{key}@{n}'] = {'hbm_bytes_per_launch': sum(vals) / len(vals), 'dispatches': len(vals), 'kernel': kname,
                                 'note': '2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate PMC passes of the same command; '
                                         'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 64 B per 128-B request)'}
json.dump(traffic, open(f'{out}/traffic.json', 'w'), indent=1)
for k, v in sorted(traffic.items()):
    print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e9:9.3f} GB x {v['dispatches']}")
