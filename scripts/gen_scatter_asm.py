"""
Generates libertem_amd/csrc/ltmi_scatter_loop.inc: the main loop of k_scatter (ltmi_scatter.hip) as
inline-asm string macros, one per pixel type.  Run from the repo root:

    python scripts/gen_scatter_asm.py

Why generated: the loop is software pipelined over two register sets (A / B) and four bundles per
block, every register is named by hand (the accumulators are addressed through the VGPR index mode,
s_set_gpr_idx_on, which the compiler cannot express), and the five pixel types differ in two
instructions.  The register map is the contract with ltmi_scatter.hip (see the comment there).
"""
import os

PITCH = 1028            # LDS bytes per frame row (1 KiB of pixels + 4: odd dword pitch, conflict-free columns)
BUF = 64 * PITCH        # one frame buffer: 64 rows
ACC0 = 56               # accumulators v[56:119], padding slots v[120:127]

H = {'A': 12, 'B': 16}              # header SGPRs s[12:15] / s[16:19]
W = {'A': 36, 'B': 68}              # weights s[36:67] / s[68:99]
X = {'A': 40, 'B': 48}              # x values v40,42,44,46 / v48,50,52,54 (odd partners: address temps)

TYPES = {
    'u8': ('ds_read_u8', 'v_cvt_f32_ubyte0'),
    'i8': ('ds_read_i8', 'v_cvt_f32_i32'),
    'u16': ('ds_read_u16', 'v_cvt_f32_u32'),
    'i16': ('ds_read_i16', 'v_cvt_f32_i32'),
    'f32': ('ds_read_b32', None),
}


ABL = set()          # timing-only ablations (u16 variants _a1.._a4): 'hotw', 'nodma', 'nolds', 'nofma', 'nobar'


def lds_reads(cur, nxt, dsread):
    """addresses + LDS reads of the NEXT block's four pixels: their row offsets are the high halves of the
    current block's header words; v33 = lane * PITCH + offset of the buffer being read"""
    out = []
    for b in range(4):
        xr, tr = X[nxt] + 2 * b, X[nxt] + 2 * b + 1
        out.append(f"v_add_u32_sdwa v{tr}, s{H[cur] + b}, v33 dst_sel:DWORD dst_unused:UNUSED_PAD "
                   f"src0_sel:WORD_1 src1_sel:DWORD")
        if 'nolds' not in ABL:
            out.append(f"{dsread} v{xr}, v{tr}")
    return out


def fmas(cur, cvt):
    """the block's 4 x 4 packed FMAs: conversions first, then ONE index-mode window -- s_set_gpr_idx_on /
    _idx take the accumulator slot from the low byte of the bundle's header word (M0[7:0]); destination and
    src2 of the v_pk_fma_f32 are relative to it (mode 0xc), the x operand and the weights are not"""
    out = []
    if cvt:
        for b in range(4):
            xr = X[cur] + 2 * b
            out.append(f"{cvt} v{xr}, v{xr}")
    for b in range(4):
        xr = X[cur] + 2 * b
        out.append(f"s_set_gpr_idx_on s{H[cur] + b}, 0xc" if b == 0 else f"s_set_gpr_idx_idx s{H[cur] + b}")
        for k in range(4):
            w = W[cur] + 8 * b + 2 * k
            a = ACC0 + 2 * k
            if 'nofma' in ABL:
                continue
            out.append(f"v_pk_fma_f32 v[{a}:{a + 1}], v[{xr}:{xr + 1}], s[{w}:{w + 1}], v[{a}:{a + 1}] "
                       f"op_sel_hi:[0,1,1]")
    out.append("s_set_gpr_idx_off")
    return out


def dma_issue(tag):
    """4 LDS-DMA instructions: this wave's rows 4j .. 4j + 3 of the chunk whose lane offsets are in v34, into
    buffer s35; row pointers in lanes 0 .. 3 of v36 / v37; s28 = LDS address of the wave's first row"""
    out = ["v_readlane_b32 s28, v32, 9", f"s_mul_i32 s29, s35, {BUF}", "s_add_u32 s28, s28, s29"]
    for k in range(4):
        out += [f"v_readlane_b32 s26, v36, {k}", f"v_readlane_b32 s27, v37, {k}",
                "s_mov_b32 m0, s28", f"s_add_u32 s28, s28, {PITCH}", "s_nop 3"]
        if 'nodma' not in ABL:
            out += ["global_load_lds_dwordx4 v34, s[26:27]"]
    return out


def table_load(ahead):
    """lane offsets of chunk s33 + ahead (clamped to the last one) -> v34"""
    return [f"s_add_u32 s28, s33, {ahead}", "s_sub_u32 s29, s34, 1", "s_min_u32 s28, s28, s29",
            "s_lshl_b32 s28, s28, 8", "v_add_u32 v39, s28, v35", "global_load_dword v34, v39, s[24:25]"]


def chunk_end(tag):
    out = ["s_waitcnt vmcnt(0)"] + ([] if 'nobar' in ABL else ["s_barrier"]) + ["v_swap_b32 v33, v38",
           "s_add_u32 s28, s33, 2", "s_cmp_lt_u32 s28, s34", f"s_cbranch_scc0 L_nodma_{tag}_%="]
    out += dma_issue(tag)
    out += table_load(3)
    out += [f"L_nodma_{tag}_%=:", "s_xor_b32 s35, s35, 1", "s_add_u32 s33, s33, 1"]
    return out


def half(cur, nxt, dsread, cvt):
    hw, ww = H[nxt], W[nxt]
    out = [f"L_{cur}_%=:", "s_waitcnt lgkmcnt(0)", f"L_{cur}_entry_%=:",
           f"s_load_dwordx4 s[{hw}:{hw + 3}], s[20:21], s30",
           f"s_load_dwordx16 s[{ww}:{ww + 15}], s[22:23], s31",
           f"s_load_dwordx16 s[{ww + 16}:{ww + 31}], s[22:23], s31 offset:0x40",
           "s_add_u32 s30, s30, 16", "s_add_u32 s31, s31, 0" if 'hotw' in ABL else "s_add_u32 s31, s31, 0x80",
           f"s_bitcmp1_b32 s{H[cur] + 3}, 8", f"s_cbranch_scc1 L_{cur}_fma_%="]
    out += lds_reads(cur, nxt, dsread)
    out += [f"L_{cur}_fma_%=:"]
    out += fmas(cur, cvt)
    out += [f"s_bitcmp1_b32 s{H[cur] + 3}, 8", f"s_cbranch_scc0 L_{cur}_next_%="]
    out += chunk_end(cur)
    out += ["s_cmp_eq_u32 s32, 1", "s_cbranch_scc1 L_done_%="]
    out += lds_reads(cur, nxt, dsread)
    out += [f"L_{cur}_next_%=:", "s_sub_u32 s32, s32, 1", "s_cmp_eq_u32 s32, 0", "s_cbranch_scc1 L_done_%="]
    return out


def loop(dsread, cvt):
    out = []
    # operands -> fixed registers
    out += ["v_mov_b32 v32, %0", "v_mov_b32 v33, %1", f"v_add_u32 v38, {BUF}, v33", "v_mov_b32 v35, %2",
            "v_mov_b32 v36, %3", "v_mov_b32 v37, %4"]
    for i, s in enumerate((20, 21, 22, 23, 24, 25, 32, 33, 34)):
        out.append(f"v_readlane_b32 s{s}, v32, {i}")
    out += [f"v_mov_b32 v{r}, 0" for r in range(ACC0, 128)]
    out += ["s_mov_b32 s30, 0", "s_mov_b32 s31, 0", "s_mov_b32 s35, 0", "s_nop 4"]
    # frame copies of the first two chunks
    out += ["s_lshl_b32 s28, s33, 8", "v_add_u32 v39, s28, v35", "global_load_dword v34, v39, s[24:25]",
            "s_waitcnt vmcnt(0)"]
    out += dma_issue('p0')
    out += table_load(1) + ["s_waitcnt vmcnt(0)"]
    out += ["s_add_u32 s28, s33, 1", "s_cmp_lt_u32 s28, s34", "s_cbranch_scc0 L_one_%=", "s_mov_b32 s35, 1"]
    out += dma_issue('p1')
    out += ["s_mov_b32 s35, 0"]
    out += table_load(2)
    out += ["L_one_%=:", "s_waitcnt vmcnt(0)", "s_barrier"]
    # pipeline prologue: block 0 (a dummy that carries the pixel offsets of block 1)
    out += [f"s_load_dwordx4 s[{H['A']}:{H['A'] + 3}], s[20:21], s30",
            f"s_load_dwordx16 s[{W['A']}:{W['A'] + 15}], s[22:23], s31",
            f"s_load_dwordx16 s[{W['A'] + 16}:{W['A'] + 31}], s[22:23], s31 offset:0x40",
            "s_add_u32 s30, s30, 16", "s_add_u32 s31, s31, 0x80", "s_waitcnt lgkmcnt(0)",
            "s_branch L_A_entry_%="]
    out += half('A', 'B', dsread, cvt)
    out += half('B', 'A', dsread, cvt)
    out += ["s_branch L_A_%=", "L_done_%=:", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)"]
    return out


def render():
    """the text of libertem_amd/csrc/ltmi_scatter_loop.inc (libertem_amd/build.py compares the tracked file with it)"""
    out = ["// GENERATED by scripts/gen_scatter_asm.py -- do not edit.  Main loop of k_scatter per pixel type;\n"
           "// operands: %0 argument lanes, %1 lane * PITCH, %2 lane * 4, %3 / %4 row pointers (lanes 0..3).\n",
           f"#define SCAT_PITCH {PITCH}\n#define SCAT_BUF {BUF}\n#define SCAT_ACC0 {ACC0}\n"]
    variants = [(name, dsread, cvt, ()) for name, (dsread, cvt) in TYPES.items()]
    for i, abl in enumerate((('hotw',), ('nodma',), ('nolds',), ('nofma',), ('nodma', 'hotw'), ('nobar',), ('nobar', 'hotw'))):
        variants.append((f'u16_a{i + 1}',) + TYPES['u16'] + (abl,))
    for name, dsread, cvt, abl in variants:
        ABL.clear()
        ABL.update(abl)
        out.append(f"#define SCAT_LOOP_{name} \\\n")
        lines = loop(dsread, cvt)
        out.append(" \\\n".join(f'    "{ln}\\n\\t"' for ln in lines))
        out.append("\n")
    ABL.clear()
    return "".join(out)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'libertem_amd', 'csrc', 'ltmi_scatter_loop.inc')
    with open(path, 'w') as f:
        f.write(render())
    print(path)


if __name__ == '__main__':
    main()
