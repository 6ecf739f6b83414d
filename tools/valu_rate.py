#!/usr/bin/env python3
"""valu_rate.py — builds and runs tools/valu_rate.hip on the GPU box and writes the table kept under profiles/.

    gpurun -- python3 tools/valu_rate.py            # -> gpurun_out/valu_rate.txt, gpurun_out/valu_rate.json

Answers VERDICT r02 item 2: what does one wave64 vector instruction of each class cost on gfx950 at 1 / 2 / 4 waves
per SIMD -- fp64 (fma / mul / add / compare / floor / convert) against f32 and 24-bit / 32-bit integer -- i.e. what an f32
or fixed-point screen in front of the exact fp64 predicates of the ray march could save.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    exe = '/tmp/valu_rate'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-Wno-unused-value', '-o', exe,
                           os.path.join(ROOT, 'tools', 'valu_rate.hip')])
    txt = subprocess.check_output([exe]).decode()
    js = json.loads(subprocess.check_output([exe, '--json']).decode())
    by = {c['kernel']: c for c in js['cases']}

    def ratio(a, b, w):
        return by[a]['clocks_per_inst_simd'][w] / by[b]['clocks_per_inst_simd'][w]

    lines = [txt, '',
             'fp64 / f32 issue cost per SIMD (independent streams): '
             + ', '.join(f'{w} waves/SIMD: fma {ratio("k_fma_f64_i", "k_fma_f32_i", w):.2f}x, '
                         f'mul {ratio("k_mul_f64_i", "k_mul_f32_i", w):.2f}x, add {ratio("k_add_f64_i", "k_add_f32_i", w):.2f}x, '
                         f'cmp {ratio("k_cmp_f64_i", "k_cmp_f32_i", w):.2f}x' for w in ('1', '2', '4')),
             'fp64 fma / 24-bit integer multiply per SIMD: '
             + ', '.join(f'{w} waves/SIMD: {ratio("k_fma_f64_i", "k_mul_u24_i", w):.2f}x' for w in ('1', '2', '4'))]
    report = '\n'.join(lines) + '\n'
    open(os.path.join(out_dir, 'valu_rate.txt'), 'w').write(report)
    json.dump(js, open(os.path.join(out_dir, 'valu_rate.json'), 'w'), indent=1)
    sys.stdout.write(report)


if __name__ == '__main__':
    main()
