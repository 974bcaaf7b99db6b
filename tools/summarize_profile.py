"""Turns what tools/profile_gpu.sh left under gpurun_out/prof_TAG/ into the files committed under profiles/.

    python tools/summarize_profile.py TAG [ROUND_DIR]

Writes profiles/ROUND_DIR/TAG_kernel_stats.csv   rocprofv3's own --stats summary (copied as is)
       profiles/ROUND_DIR/TAG_by_launch.csv      rt2d_* dispatches grouped by (kernel, grid size): the 999-ray fan and the
                                                 1M-ray fan launch the same kernels, the grid size tells them apart
       profiles/ROUND_DIR/TAG_traffic.json       HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
                                                 MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950
                                                 FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled
                                                 (calibrated there on wide coalesced streaming reads); WRITE_SIZE is exact
                                                 for streaming stores and atomics.
       profiles/ROUND_DIR/TAG_counters.json      (when the valu / atomic passes ran) per launch: vector and scalar
                                                 instructions issued, the FP64 op mix, waves, LDS instructions, L2
                                                 atomic requests; with the launch duration of the trace pass: FP64 flop/s
                                                 as a fraction of the 78.6 TFLOP/s vector peak (a wave instruction
                                                 counts 64 lanes, FMA twice: an upper bound where lanes are masked off)
                                                 and atomics per second.
bench.py reads TAG_traffic.json of the newest committed profile for `roofline.traffic`.
"""

import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*\)$', '', name)


def main():
    tag = sys.argv[1]
    rdir = sys.argv[2] if len(sys.argv) > 2 else 'r01'
    src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
    dst = os.path.join(ROOT, 'profiles', rdir)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))

    groups = defaultdict(list)
    with open(os.path.join(src, 'trace', 't_kernel_trace.csv')) as f:
        for row in csv.DictReader(f):
            if 'rt2d_' not in row['Kernel_Name']:
                continue
            grid = int(row['Grid_Size_X']) * int(row['Grid_Size_Y']) * int(row['Grid_Size_Z'])
            key = (short(row['Kernel_Name']), grid, int(row['Workgroup_Size_X']))
            groups[key].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    # which build was measured: written on the GPU box next to the passes (tools/profile_gpu.sh)
    build = None
    bpath = os.path.join(src, 'build.json')
    if os.path.exists(bpath):
        with open(bpath) as f:
            build = json.load(f)
    pmc = {}
    for which, sub, fname in (('FETCH_SIZE', 'fetch', 'f_counter_collection.csv'), ('WRITE_SIZE', 'write', 'w_counter_collection.csv')):
        path = os.path.join(src, sub, fname)
        acc = defaultdict(list)
        if os.path.exists(path):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if 'rt2d_' not in row['Kernel_Name'] or row['Counter_Name'] != which:
                        continue
                    acc[(short(row['Kernel_Name']), int(row['Grid_Size']), int(row['Workgroup_Size']))].append(float(row['Counter_Value']))
        pmc[which] = acc

    rows = []
    traffic = {}
    for key in sorted(groups, key=lambda k: (k[0], k[1])):
        d = groups[key]
        fetch = pmc['FETCH_SIZE'].get(key)
        write = pmc['WRITE_SIZE'].get(key)
        fetch_b = None if not fetch else 2 * 1024 * sum(fetch) / len(fetch)          # gfx950 correction: x2
        write_b = None if not write else 1024 * sum(write) / len(write)
        rows.append(dict(kernel=key[0], grid_threads=key[1], workgroup=key[2], launches=len(d), avg_ns=sum(d) / len(d),
                         min_ns=min(d), max_ns=max(d),
                         hbm_read_bytes='' if fetch_b is None else round(fetch_b),
                         hbm_write_bytes='' if write_b is None else round(write_b)))
        traffic[f'{key[0]}|grid={key[1]}|wg={key[2]}'] = dict(
            avg_ns=sum(d) / len(d), launches=len(d), hbm_read_bytes=fetch_b, hbm_write_bytes=write_b,
            hbm_bytes=None if fetch_b is None or write_b is None else fetch_b + write_b)
    with open(os.path.join(dst, f'{tag}_by_launch.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    with open(os.path.join(dst, f'{tag}_traffic.json'), 'w') as f:
        json.dump(dict(units='bytes per launch; FETCH_SIZE [KiB] x 1024 x 2 (gfx950), WRITE_SIZE [KiB] x 1024',
                       build=build, kernels=traffic), f, indent=1)
    # optional passes: instruction mix and L2 atomics
    names = ('SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64',
             'SQ_INSTS_VALU_TRANS_F64', 'SQ_INSTS_SALU', 'SQ_WAVES', 'SQ_INSTS_LDS', 'TCC_ATOMIC_sum', 'TCC_REQ_sum')
    cnt = defaultdict(lambda: defaultdict(list))
    for sub, fname in (('valu', 'v_counter_collection.csv'), ('atomic', 'a_counter_collection.csv')):
        path = os.path.join(src, sub, fname)
        if os.path.exists(path):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if 'rt2d_' in row['Kernel_Name'] and row['Counter_Name'] in names:
                        key = (short(row['Kernel_Name']), int(row['Grid_Size']), int(row['Workgroup_Size']))
                        cnt[key][row['Counter_Name']].append(float(row['Counter_Value']))
    if cnt:
        out = {}
        for key in sorted(cnt, key=lambda k: (k[0], k[1])):
            c = {nm: sum(v) / len(v) for nm, v in cnt[key].items()}
            d = groups.get(key)
            if d:
                t = sum(d) / len(d) * 1e-9
                c['avg_ns'] = sum(d) / len(d)
                if 'SQ_INSTS_VALU_FMA_F64' in c:
                    flop = 64 * (2 * c['SQ_INSTS_VALU_FMA_F64'] + c.get('SQ_INSTS_VALU_MUL_F64', 0) +
                                 c.get('SQ_INSTS_VALU_ADD_F64', 0) + c.get('SQ_INSTS_VALU_TRANS_F64', 0))
                    c['fp64_flop_upper_bound'] = flop
                    c['fp64_vector_frac_of_78.6_TFLOPs'] = flop / t / 78.6e12
                if 'SQ_INSTS_VALU' in c:
                    c['valu_insts_per_s'] = c['SQ_INSTS_VALU'] / t
                    c['valu_issue_frac'] = c['SQ_INSTS_VALU'] * 4 / (t * 2.4e9 * 1024)     # 4 cycles per wave64 op, 1024 SIMDs, 2.4 GHz
                if 'TCC_ATOMIC_sum' in c:
                    c['atomics_per_s'] = c['TCC_ATOMIC_sum'] / t
            out[f'{key[0]}|grid={key[1]}|wg={key[2]}'] = c
        with open(os.path.join(dst, f'{tag}_counters.json'), 'w') as f:
            json.dump(dict(units='counter values per launch (mean over the launches of the pass)', build=build, kernels=out), f, indent=1)
    for line in open(os.path.join(src, 'trace.log')):
        if line.startswith('{'):
            with open(os.path.join(dst, f'{tag}_bench.json'), 'w') as f:
                f.write(line)
    for r in rows:
        print(f"{r['kernel'][:60]:60s} grid {r['grid_threads']:>9d} x{r['launches']:<3d} avg {r['avg_ns'] / 1e3:10.1f} us  "
              f"rd {r['hbm_read_bytes'] or 0:>12} wr {r['hbm_write_bytes'] or 0:>12}")


if __name__ == '__main__':
    main()
