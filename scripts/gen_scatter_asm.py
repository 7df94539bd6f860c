"""
Generates libertem_amd/csrc/ltmi_scatter_loop.inc: the main loop of k_scatter (ltmi_scatter.hip) as
inline-asm string macros, one per pixel type (+ timing-only ablations of the uint16 one).  Run from the
repo root:

    python scripts/gen_scatter_asm.py

Why generated: the loop is software pipelined over two register sets (A / B) and four bundles per
block, every register is named by hand (the accumulators are addressed through the VGPR index mode,
s_set_gpr_idx_on, which the compiler cannot express), and the five pixel types differ in two
instructions.  The register map is the contract with ltmi_scatter.hip (see the comment there).

Frames in LDS: 4 buffers of one CHUNK = 512 bytes of each of the 64 frame rows; two frames share a
1028-byte LDS row (frame 2k at +0, frame 2k + 1 at +512: one global_load_lds_dwordx4 fills it, lanes
0-31 from one frame, 32-63 from the next).  The copy of chunk c + 4 starts when chunk c has been left by
every wave, so three chunks are in flight while one is read.
"""
import os

PITCH = 1028            # LDS bytes per pair of frame rows (2 x 512 + 4: odd dword pitch)
ROWB = 512              # bytes of a frame row per chunk
BUF = 32 * PITCH        # one frame buffer: 64 rows
NBUF = 4
ACC0 = 56               # accumulators v[56:119], padding slots v[120:127]

H = {'A': 12, 'B': 20}              # header SGPRs s[12:19] / s[20:27]: 4 bundle words + 4 segment offsets
W = {'A': 36, 'B': 68}              # weights s[36:67] / s[68:99]
X = {'A': 40, 'B': 48}              # x values v40,42,44,46 / v48,50,52,54 (odd partners: address temps)
# s[28:29] header stream, s[30:31] weight stream, s32 header offset, s33 chunks left after the current one,
# s34 buffer being read, s35 / vcc temps
# v28 lane's LDS offset in a buffer, v29 last valid 16-byte piece of a row, v30:31 address temp, v32 argument
# lanes, v33 v28 + buffer, v34 lane's byte offset in the frame row, v35 (lane & 7) * 16, v36:37 / v38:39 row
# pointers of the wave's two copy instructions

TYPES = {
    'u8': ('ds_read_u8', 'v_cvt_f32_ubyte0'),
    'i8': ('ds_read_i8', 'v_cvt_f32_i32'),
    'u16': ('ds_read_u16', 'v_cvt_f32_u32'),
    'i16': ('ds_read_i16', 'v_cvt_f32_i32'),
    'f32': ('ds_read_b32', None),
}
ABL = set()          # timing-only ablations (u16 variants _a1 ..): 'hotw', 'nodma', 'nolds', 'nofma', 'nobar'


def lds_reads(cur, nxt, dsread):
    """addresses + LDS reads of the NEXT block's four pixels: their row offsets are the high halves of the
    current block's header words"""
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


def dma_issue(seg, bufreg):
    """the wave's two copy instructions of the chunk whose four segment offsets are s[seg : seg + 3], into
    buffer `bufreg`.  Lane l fetches the 16-byte piece (l & 7) of segment (l & 31) >> 3; pieces behind the end
    of the row (v29) fetch offset 0 instead (never read)."""
    out = [f"v_mov_b32 v34, s{seg}"]
    for g, m in ((1, '0xff00'), (2, '0xff0000'), (3, '0xff000000')):
        out += [f"v_mov_b32 v30, s{seg + g}", f"s_mov_b32 vcc_lo, {m}", f"s_mov_b32 vcc_hi, {m}",
                "v_cndmask_b32 v34, v34, v30, vcc"]
    out += ["v_add_u32 v34, v34, v35", "v_cmp_gt_u32 vcc, v34, v29", "v_cndmask_b32 v34, v34, 0, vcc",
            "v_readlane_b32 s35, v32, 5", f"s_mul_i32 vcc_lo, {bufreg}, {BUF}", "s_add_u32 s35, s35, vcc_lo"]
    for k, rp in ((0, 36), (1, 38)):
        out += [f"v_add_co_u32 v30, vcc, v{rp}, v34", f"v_addc_co_u32 v31, vcc, 0, v{rp + 1}, vcc",
                "s_mov_b32 m0, s35", f"s_add_u32 s35, s35, {PITCH}", "s_nop 0"]
        if 'nodma' not in ABL:
            out += ["global_load_lds_dwordx4 v[30:31], off"]
    return out


def set_read_buffer():
    return [f"s_mul_i32 s35, s34, {BUF}", "v_add_u32 v33, s35, v28"]


def chunk_end(cur, tag):
    """the wave has finished chunk c: its share of chunk c + 1 must have landed (two later copies may be
    in flight), everybody meets, the copy of chunk c + 4 goes into the buffer just left"""
    out = ["s_cmp_ge_u32 s33, 3", f"s_cbranch_scc0 L_w2_{tag}_%=", "s_waitcnt vmcnt(4)", f"s_branch L_wd_{tag}_%=",
           f"L_w2_{tag}_%=:", "s_cmp_eq_u32 s33, 2", f"s_cbranch_scc0 L_w0_{tag}_%=", "s_waitcnt vmcnt(2)",
           f"s_branch L_wd_{tag}_%=", f"L_w0_{tag}_%=:", "s_waitcnt vmcnt(0)", f"L_wd_{tag}_%=:"]
    if 'nobar' not in ABL:
        out += ["s_barrier"]
    out += ["s_cmp_ge_u32 s33, 4", f"s_cbranch_scc0 L_nodma_{tag}_%="]
    out += dma_issue(H[cur] + 4, 's34')
    out += [f"L_nodma_{tag}_%=:", "s_add_u32 s34, s34, 1", "s_and_b32 s34, s34, 3"]
    out += set_read_buffer()
    return out


def half(cur, nxt, dsread, cvt):
    hw, ww = H[nxt], W[nxt]
    out = [f"L_{cur}_%=:", "s_waitcnt lgkmcnt(0)", f"L_{cur}_entry_%=:",
           f"s_load_dwordx8 s[{hw}:{hw + 7}], s[28:29], s32",
           "s_lshl_b32 s35, s32, 2"]
    if 'hotw' in ABL:
        out += ["s_and_b32 s35, s35, 0x780"]
    out += [f"s_load_dwordx16 s[{ww}:{ww + 15}], s[30:31], s35",
            f"s_load_dwordx16 s[{ww + 16}:{ww + 31}], s[30:31], s35 offset:0x40",
            "s_add_u32 s32, s32, 32",
            f"s_bitcmp1_b32 s{H[cur] + 3}, 8", f"s_cbranch_scc1 L_{cur}_fma_%="]
    out += lds_reads(cur, nxt, dsread)
    out += [f"L_{cur}_fma_%=:"]
    out += fmas(cur, cvt)
    out += [f"s_bitcmp1_b32 s{H[cur] + 3}, 8", f"s_cbranch_scc0 L_{cur}_next_%="]
    out += chunk_end(cur, cur)
    out += ["s_cmp_eq_u32 s33, 0", "s_cbranch_scc1 L_done_%=", "s_sub_u32 s33, s33, 1"]
    out += lds_reads(cur, nxt, dsread)
    out += [f"L_{cur}_next_%=:"]
    return out


def loop(dsread, cvt):
    out = []
    # operands -> fixed registers
    out += ["v_mov_b32 v32, %0", "v_mov_b32 v28, %1", "v_mov_b32 v35, %2",
            "v_mov_b32 v36, %3", "v_mov_b32 v37, %4", "v_mov_b32 v38, %5", "v_mov_b32 v39, %6",
            "v_mov_b32 v29, %7"]
    for i, s in enumerate((28, 29, 30, 31, 33)):
        out.append(f"v_readlane_b32 s{s}, v32, {i}")
    out += ["v_readlane_b32 s12, v32, 6", "v_readlane_b32 s13, v32, 7"]
    out += [f"v_mov_b32 v{r}, 0" for r in range(ACC0, 128)]
    out += ["s_mov_b32 s32, 0", "s_mov_b32 s34, 0", "s_nop 4"]
    # workgroups start out of phase (the chip would otherwise ask for its chunks in bursts)
    out += ["v_readlane_b32 s35, v32, 8", "L_ph_%=:", "s_cmp_eq_u32 s35, 0", "s_cbranch_scc1 L_ph1_%=",
            "s_sleep 8", "s_sub_u32 s35, s35, 1", "s_branch L_ph_%=", "L_ph1_%=:"]
    # frame copies of the first (up to) four chunks: their segment offsets from the pass's table
    out += [f"s_load_dwordx16 s[{W['A']}:{W['A'] + 15}], s[12:13], 0", "s_waitcnt lgkmcnt(0)"]
    for c in range(NBUF):
        if c:
            out += [f"s_cmp_ge_u32 s33, {c}", "s_cbranch_scc0 L_pro_%="]
        out += [f"s_mov_b32 s34, {c}"] + dma_issue(W['A'] + 4 * c, 's34')
    out += ["L_pro_%=:", "s_mov_b32 s34, 0"] + set_read_buffer()
    # chunk 0 must have landed: it is the oldest of up to four shares
    out += ["s_waitcnt vmcnt(0)", "s_barrier"]
    # pipeline prologue: block 0 (a dummy that carries the pixel offsets of block 1)
    out += [f"s_load_dwordx8 s[{H['A']}:{H['A'] + 7}], s[28:29], s32",
            f"s_load_dwordx16 s[{W['A']}:{W['A'] + 15}], s[30:31], 0",
            f"s_load_dwordx16 s[{W['A'] + 16}:{W['A'] + 31}], s[30:31], 0x40",
            "s_add_u32 s32, s32, 32", "s_waitcnt lgkmcnt(0)",
            "s_branch L_A_entry_%="]
    out += half('A', 'B', dsread, cvt)
    out += half('B', 'A', dsread, cvt)
    out += ["s_branch L_A_%=", "L_done_%=:", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)"]
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'libertem_amd', 'csrc', 'ltmi_scatter_loop.inc')
    with open(path, 'w') as f:
        f.write("// GENERATED by scripts/gen_scatter_asm.py -- do not edit.  Main loop of k_scatter per pixel type;\n"
                "// operands: %0 argument lanes, %1 lane's LDS offset in a buffer, %2 (lane & 7) * 16, %3 .. %6 row\n"
                "// pointers (lo, hi) of the wave's two copy instructions, %7 last valid piece offset of a row.\n")
        f.write(f"#define SCAT_PITCH {PITCH}\n#define SCAT_ROWB {ROWB}\n#define SCAT_BUF {BUF}\n"
                f"#define SCAT_NBUF {NBUF}\n#define SCAT_ACC0 {ACC0}\n")
        variants = [(name, dsread, cvt, ()) for name, (dsread, cvt) in TYPES.items()]
        for i, abl in enumerate((('hotw',), ('nodma',), ('nolds',), ('nofma',), ('nodma', 'hotw'), ('nobar',))):
            variants.append((f'u16_a{i + 1}',) + TYPES['u16'] + (abl,))
        for name, dsread, cvt, abl in variants:
            ABL.clear()
            ABL.update(abl)
            f.write(f"#define SCAT_LOOP_{name} \\\n")
            lines = loop(dsread, cvt)
            f.write(" \\\n".join(f'    "{ln}\\n\\t"' for ln in lines))
            f.write("\n")
        ABL.clear()
    print(path)


if __name__ == '__main__':
    main()
