#!/bin/bash
# Developer tool (GPU box): SQ issue counters of chosen kernels, one counter set per rocprofv3 pass (--pmc with --kernel-trace
# only) -> gpurun_out/pmc_sq_<tag>/summary.txt (per-launch medians per kernel).
# usage: tools/pmc_sq.sh <tag> <kernel-name-substrings, comma separated> -- <python program and arguments>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; KERNELS=$2; shift 3
O=gpurun_out/pmc_sq_$TAG; rm -rf $O; mkdir -p $O
pass() { local tag=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $O/$tag -o pmc -- "$@" > $O/$tag.log 2>&1
  cp "$(find $O/$tag -name '*counter_collection.csv' | head -1)" $O/${tag}.csv && rm -rf $O/$tag; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- "$@"
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- "$@"
pass c SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVES -- "$@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- "$@" > $O/kt.log 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv; rm -rf $O/kt
KERNELS="$KERNELS" OUT="$O" python3 - <<'PY' > $O/summary.txt
import csv, collections, statistics, glob, os
names = os.environ['KERNELS'].split(','); O = os.environ['OUT']
out = collections.defaultdict(dict)
def short(k):
    return k.replace('(anonymous namespace)::', '').replace('void ', '').split('<')[0].split('(')[0]
for f in sorted(glob.glob(O + '/[abc].csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if any(n in k for n in names): acc[short(k)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items(): out[k][c] = statistics.median(v)
for r in csv.DictReader(open(O + '/kernel_stats.csv')):
    k = short(r['Name'])
    if k in out: out[k]['avg_us'] = float(r['AverageNs']) / 1e3
for k, cs in out.items():
    print(k)
    for c in sorted(cs): print("   %-28s %14.1f" % (c, cs[c]))
    if 'SQ_BUSY_CYCLES' in cs and 'SQ_ACTIVE_INST_ANY' in cs:
        cap = cs['SQ_BUSY_CYCLES'] / 32 / 4 * 1024          # issue slots of the launch: busy cycles per shader engine / 4, times 1024 SIMDs
        print("   %-28s %14.3f" % ('issue_slots_used_frac', cs['SQ_ACTIVE_INST_ANY'] / cap))
        print("   %-28s %14.3f" % ('valu_frac_of_slots', cs.get('SQ_ACTIVE_INST_VALU', 0) / cap))
PY
cat $O/summary.txt
