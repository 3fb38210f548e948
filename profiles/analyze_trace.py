"""Group one forward's dispatches of a rocprofv3 --kernel-trace CSV by (kernel class, grid)."""
import collections
import csv
import sys


def short(n):
    if 'gemm_kernel' in n:
        return 'conv' if ('Li1EEE' in n or 'Accum' in n) else 'gemm'
    for k in ('attn_kernel', 'gn_stats', 'gn_apply', 'ln_kernel', 'skinny', 'splitk'):
        if k in n:
            return k + (n[n.find('Li'):n.find('EEE')] if k == 'attn_kernel' else '')
    return n[:24]


rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if 'prep_image' in r['Kernel_Name']]
s, e = idx[-4], idx[-3]
fw = rows[s:e]
print(len(fw), 'dispatches in one forward')
tot = 0
groups = collections.OrderedDict()
for r in fw:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
    key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    g = groups.setdefault(key, [0, 0.0]); g[0] += 1; g[1] += d; tot += d
for k, v in sorted(groups.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(k, v[0], round(v[1], 1), 'us total', round(v[1] / v[0], 1), 'us each')
print('sum kernel us', round(tot, 1), ' span us', (int(fw[-1]['End_Timestamp']) - int(fw[0]['Start_Timestamp'])) / 1000)
