#!/usr/bin/env python3
"""isa_stats.py [lib.so] [name-filter ...] — static instruction mix of the gfx950 code in libd2d_hip.so, per kernel / device function.

A counted ISA walk: vector instructions by class (f64 / f32 / integer + moves / compares / conversions /
DPP + lane ops), scalar instructions, LDS, global / scratch memory, lane spills (v_readlane / v_writelane), waits, branches, and the
register / scratch figures of the kernel descriptor notes.  No GPU needed (llvm-objdump of the embedded code object).
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    op = re.sub(r'_(e32|e64|dpp|sdwa|e64_dpp)$', '', op)
    if op.startswith('v_'):
        if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
            return 'lane'
        if op.startswith('v_cmp') or op.startswith('v_cmpx'):
            return 'vcmp_f64' if op.endswith('f64') else 'vcmp'
        if op.startswith('v_cvt'):
            return 'vcvt_f64' if 'f64' in op else 'vcvt'
        if op.endswith('_f64') or '_f64_' in op:
            return 'v_f64'
        if op.endswith(('_f32', '_f16')) or '_f32_' in op:
            return 'v_f32'
        if op.startswith(('v_mov', 'v_cndmask', 'v_accvgpr')):
            return 'v_mov'
        if op.endswith(('_dpp',)):
            return 'v_int'
        return 'v_int'
    if op.startswith('s_'):
        if op.startswith('s_waitcnt'):
            return 's_wait'
        if op.startswith(('s_cbranch', 's_branch', 's_call', 's_setpc', 's_swappc', 's_getpc')):
            return 's_branch'
        if op.startswith(('s_load', 's_buffer_load')):
            return 's_load'
        if op.startswith(('s_nop', 's_sleep', 's_setprio', 's_barrier')):
            return 's_misc'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('scratch_'):
        return 'scratch'
    if op.startswith(('global_', 'flat_', 'buffer_')):
        return 'vmem'
    return 'other'


def main():
    lib = os.path.join(ROOT, 'gym-drone2d-activeperception_amd', 'csrc', 'libd2d_hip.so')
    args = sys.argv[1:]
    if args and args[0].endswith('.so'):
        lib = args.pop(0)
    filt = args
    tmp = tempfile.mkdtemp(prefix='isa_')
    try:
        local = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, local)
        subprocess.check_call([f'{LLVM}/llvm-objdump', '--offloading', local], stdout=subprocess.DEVNULL, cwd=tmp)
        co = [f for f in os.listdir(tmp) if 'gfx950' in f]
        if not co:
            sys.exit('no gfx950 code object in ' + lib)
        co = os.path.join(tmp, co[0])
        dis = subprocess.check_output([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', '-C', co], text=True)
        notes = subprocess.check_output([f'{LLVM}/llvm-readelf', '--notes', co], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # kernel metadata: name -> vgpr / sgpr / spills / scratch
    meta = {}
    cur = {}
    for ln in notes.splitlines():
        m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == 'agpr_count' and cur.get('name'):
            pass
        if k in ('name', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
                 'group_segment_fixed_size', 'agpr_count'):
            if k == 'name' and 'symbol' not in cur and cur.get('name') and cur.get('vgpr_count') is not None:
                meta[cur['name']] = cur
                cur = {}
            if k == 'name' and v.startswith('_Z'):
                cur = {'name': v}
            elif k != 'name':
                cur[k] = v
    if cur.get('name'):
        meta[cur['name']] = cur
    funcs = collections.OrderedDict()
    name = None
    for ln in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', ln)
        if m:
            name = m.group(1)
            funcs[name] = collections.Counter()
            continue
        m = re.match(r'^\s+([a-z_0-9]+)\b', ln)
        if m and name:
            funcs[name][classify(m.group(1))] += 1
            funcs[name]['_total'] += 1
    cols = ['v_f64', 'vcmp_f64', 'vcvt_f64', 'v_f32', 'v_int', 'v_mov', 'vcmp', 'vcvt', 'lane', 'salu', 's_load', 's_wait', 's_branch',
            'lds', 'vmem', 'scratch']
    print(f'{"function":58s} {"total":>6s} {"VALU":>6s} ' + ' '.join(f'{c:>8s}' for c in cols))
    for fn, c in funcs.items():
        if c['_total'] < 50:
            continue
        if filt and not any(f in fn for f in filt):
            continue
        valu = sum(c[k] for k in ('v_f64', 'vcmp_f64', 'vcvt_f64', 'v_f32', 'v_int', 'v_mov', 'vcmp', 'vcvt', 'lane'))
        short = re.sub(r'\(anonymous namespace\)::', '', fn)
        short = re.sub(r'\(.*', '', short)[:58]
        print(f'{short:58s} {c["_total"]:6d} {valu:6d} ' + ' '.join(f'{c[k]:8d}' for k in cols))
    print()
    for n, m in meta.items():
        dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem)
        dem = re.sub(r'\(.*', '', dem)
        if filt and not any(f in dem for f in filt):
            continue
        print(f'{dem[:58]:58s} vgpr {m.get("vgpr_count", "?"):>4s} agpr {m.get("agpr_count", "?"):>3s} sgpr {m.get("sgpr_count", "?"):>4s} '
              f'vgpr_spill {m.get("vgpr_spill_count", "?"):>4s} sgpr_spill {m.get("sgpr_spill_count", "?"):>4s} '
              f'scratch {m.get("private_segment_fixed_size", "?"):>6s} B')


if __name__ == '__main__':
    main()
