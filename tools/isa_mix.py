"""tools/isa_mix.py <file.s> [regex] [--ops]: static instruction mix per kernel of a device-only assembly listing (tools/devasm.sh writes
/tmp/dis/<name>.s): FMAs, adds, cross-lane moves, other vector, LDS, memory, scalar, waits; registers / scratch / occupancy.
"""
import re, sys, subprocess, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None; counts = {}; meta = {}
names = {}
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); counts[cur] = collections.Counter(); continue
    if cur is None: continue
    s = line.strip()
    if s.startswith('.end_amdhsa_kernel') : pass
    m2 = re.match(r'^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte): (\d+)', s)
    if m2: meta.setdefault(cur, {})[m2.group(1)] = int(m2.group(2)); continue
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
    op = s.split()[0]
    counts[cur][op] += 1
def cat(op):
    if op.startswith('v_fma_f64') or op.startswith('v_mul_f64') or op.startswith('v_pk_fma') or op.startswith('v_fmac') or op.startswith('v_fma_f32') or op.startswith('v_pk_mul'): return 'fma'
    if op.startswith('v_add_f64') or op.startswith('v_add_f32') or op.startswith('v_pk_add'): return 'fadd'
    if 'dpp' in op or op.startswith('v_permlane') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'): return 'xlane'
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'
import subprocess
for k in counts:
    dem = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if pat and not re.search(pat, dem): continue
    c = collections.Counter()
    for op, n in counts[k].items(): c[cat(op)] += n
    tot_valu = c['fma'] + c['fadd'] + c['xlane'] + c['valu_other']
    print(dem[:110]); print('   ', meta.get(k), dict(c), 'VALU', tot_valu)
    if '--ops' in sys.argv:
        print('   ', sorted([(n,op) for op,n in counts[k].items() if cat(op) in ('valu_other','xlane')], reverse=True)[:25])
