#!/usr/bin/env python3
"""Static instruction mix of the hot basic blocks of a frame-kernel instantiation (runs here: hipcc cross-compiles, no GPU needed).

    python tools/isa_blocks.py [mangled-name-fragment, default the 2048-FFT 4x4 kernel]  [min block size]

Compiles csrc/frame_slab_kernel.hip to gfx950 assembly, splits the chosen kernel into basic blocks and prints, for every block of at
least `min` instructions, the counts by class with two issue estimates per wave from the MEASURED issue costs of gfx950
(tools/micro/valu_issue_rate.hip, profiles/r04_valu_issue_rate.txt).  A wave ALONE issues an instruction every 5.2 cycles (a
transcendental every 9.2).  A SIMD shared by four waves retires a plain 32-bit vector instruction (v_fma_f32, v_xor_b32, ...) every 2.55
cycles, a packed-f32 or 32-bit integer-multiply instruction (v_pk_fma_f32, v_mad_u64_u32, v_mul_lo/hi_u32) every 4.27 -- HALF rate, not the
quarter rate rounds 1-3 of this project assumed for the integer multiplies -- and a transcendental every 8.26.  `alone` = this wave by
itself, `shared` = its share of a saturated SIMD.  The synthesis stretch of a unit (symbol, rx group) is the block that holds the
Philox multiplies (v_mad_u64_u32) together with the packed FMAs."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frag = sys.argv[1] if len(sys.argv) > 1 else "ILi896ELi4ELi4ELi4ELi1ELi1ELb0ELb0E"
minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 150
src = os.path.join(ROOT, "channel-estimation-in-5g-network_amd", "csrc", "frame_slab_kernel.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5chest17frame_slab_kernel" + frag) and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), []]; blocks.append(cur); continue
    if cur is not None and re.match(r"\t[vsdgb][a-z_0-9]+", l):
        cur[1].append(l.split()[0])
HALF = ("v_pk_", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")
TRANS = ("v_sin_f32", "v_cos_f32", "v_log_f32", "v_sqrt_f32", "v_rcp_f32", "v_rcp_iflag", "v_exp_f32", "v_rsq_f32")
print(f"kernel frame_slab_kernel<{frag}>: {sum(len(b[1]) for b in blocks)} instructions in {len(blocks)} blocks; blocks of >= {minsz}:")
for name, ins in blocks:
    if len(ins) < minsz:
        continue
    c = lambda *p: sum(1 for x in ins if x.startswith(p))
    vec = [x for x in ins if x.startswith("v_")]
    q = sum(1 for x in vec if x.startswith(TRANS))
    h = sum(1 for x in vec if x.startswith(HALF))
    alone = 5.2 * (len(ins) - q) + 9.2 * q                   # every instruction of the wave, vector or scalar, takes an issue turn
    shared = 2.55 * (len(vec) - q - h) + 4.27 * h + 8.26 * q
    print(f"  {name:12s} total {len(ins):4d} | vector {len(vec):4d} (pk_fma {c('v_pk_fma'):3d} pk_mul {c('v_pk_mul'):2d} pk_add {c('v_pk_add'):2d} fma/mul/add f32 {c('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32'):3d} "
          f"mad_u64 {c('v_mad_u64'):2d} transcendental {c('v_sin', 'v_cos', 'v_log', 'v_sqrt', 'v_rcp'):2d} mov {c('v_mov', 'v_pk_mov', 'v_accvgpr'):2d}) scalar {c('s_'):3d} (s_load {c('s_load'):2d}) "
          f"lds {c('ds_'):2d} vmem {c('global_', 'scratch_', 'buffer_'):2d} | issue cycles per wave: ~{alone:.0f} alone, ~{shared:.0f} of a shared SIMD")
