#!/usr/bin/env python3
"""Writes the unrolled symbol blocks of the pipelined entropy kernel's decoder (cool_chic_amd/csrc/ccd_dec_block{16,32}.inc and
ccd_dec_tramp{16,32}.inc): string-literal lines that ccd_entropy_pipe.hip includes into the asm region of decoder_grid.

A block decodes a FULL batch of N symbols: the three copies of the symbol loop in rotation without index arithmetic, bound test
or branch; rows at immediate offsets from the batch's first row (v50); the range alternates between s[52:53] (even symbols) and
s[48:49] (odd).  A symbol whose new range has a zero high word leaves through its trampoline, which restores the conventions
of the 3-copy loop (i, v50, range registers) and continues in that loop's handler.  The 32-symbol block publishes its first
16 symbols half-way (ring cells + pixel count), so progress is published as often as with 16-symbol batches.
    python tools/gen_decoder_block.py"""
from pathlib import Path

PAIRS = [("v40", "v41", "v[46:47]"), ("v42", "v43", "v[40:41]"), ("v46", "v47", "v[42:43]")]  # (L, P) of the current row, where row j + 2 goes


def q(s):
    return '                "%s\\n\\t"' % s


def block(n, tramp0, mid_publish):
    out = []
    for j in range(n):
        cur_l, cur_p, nxt = PAIRS[j % 3]
        rcur = "s[52:53]" if j % 2 == 0 else "s[48:49]"
        rnew = ("s48", "s49") if j % 2 == 0 else ("s52", "s53")
        wait = "s_waitcnt lgkmcnt(4)" if (mid_publish and j in (n // 2, n // 2 + 1)) else "s_waitcnt lgkmcnt(2)"
        if mid_publish and j == n // 2:
            out += ["                // half-way: the first %d symbols go to the ring and the pixel count is published (the producers of the next" % (n // 2),
                    "                // step's first parts start now, as they would behind a %d-symbol batch)" % (n // 2),
                    q("s_mov_b64 exec, 0x%x" % ((1 << (n // 2)) - 1)), q("v_sub_u32 v52, %[top], %[raw]"), q("v_add_u32 v52, 1, v52"),
                    q("ds_write_b8 %[ring], v52"), q("s_mov_b64 exec, -1"), q("s_add_u32 s58, %[pix0], %[i]"), q("s_add_u32 s58, s58, %d" % (n // 2)),
                    q("v_mov_b32 v52, s58"), q("ds_write_b32 %[rdy], v52 offset:68")]
        out += [q("s_lshr_b64 s[40:41], %s, 24" % rcur),
                q("ds_read_b64 %s, v50 offset:%d" % (nxt, 512 * (j + 2))),
                q(wait),
                q("v_mad_u64_u32 v[44:45], s[42:43], s40, %s, 0" % cur_l),
                q("v_mad_u32_u24 v45, %s, s41, v45" % cur_l),
                q("v_cmpx_ge_u64 vcc, s[50:51], v[44:45]"),
                q("v_mad_u64_u32 v[48:49], s[42:43], s40, %s, 0" % cur_p),
                q("v_mad_u32_u24 v49, %s, s41, v49" % cur_p),
                q("s_ff1_i32_b64 s44, vcc"),
                q("v_writelane_b32 %%[raw], s44, %d" % j),
                q("v_readfirstlane_b32 %s, v48" % rnew[0]),
                q("v_readfirstlane_b32 %s, v49" % rnew[1]),
                q("v_readfirstlane_b32 s46, v44"),
                q("v_readfirstlane_b32 s47, v45"),
                q("s_mov_b64 exec, -1"),
                q("s_cmp_eq_u32 %s, 0" % rnew[1]),
                q("s_cbranch_scc1 %df" % (tramp0 + j)),
                q("s_sub_u32 s50, s50, s46"),
                q("s_subb_u32 s51, s51, s47")]
    out.append(q("s_add_u32 %%[i], %%[i], %d" % n))
    return out


def trampolines(n, tramp0):
    out = []
    for j in range(n):
        out.append(q("%d:" % (tramp0 + j)))
        if j % 2:
            out += [q("s_mov_b64 s[42:43], s[52:53]"), q("s_mov_b64 s[52:53], s[48:49]"), q("s_mov_b64 s[48:49], s[42:43]")]
        if j:
            out.append(q("s_add_u32 %%[i], %%[i], %d" % j))
        if j // 3:
            out.append(q("v_add_u32 v50, 0x%x, v50" % (0x600 * (j // 3))))
        out.append(q("s_branch %db" % (40 + j % 3)))
    return out


# ---- the 16-symbol block with ONE renormalisation / sentinel test per TWO symbols --------------------------------------------
# A not-taken s_cbranch costs ~17 ticks on this chain.  An even symbol is not tested at all: if its new range has a zero high
# word, the odd symbol behind it computes garbage from a tiny scale, and the test after that one (min of the two high words)
# sends both back: nothing of the even symbol's inputs has been overwritten by then - the range rotates through THREE register
# pairs, the distance through two, the hit lane's products and index through two sets - so the trampoline hands the even
# symbol to the loop's handler exactly as the per-symbol test would have.
RP = [("s52", "s53"), ("s48", "s49"), ("s74", "s75")]   # range pairs: symbol j reads RP[j % 3], writes RP[(j + 1) % 3]
DP = [("s50", "s51"), ("s76", "s77")]                    # distance pairs: symbol j compares DP[j % 2], writes DP[(j + 1) % 2]
HS = [("s46", "s47", "s44"), ("s78", "s79", "s80")]      # (scale * L low, high, hit lane) of even / odd symbols


def pair(p):
    return "s[%s:%s]" % (p[0][1:], p[1][1:])


def block_paired(n, tramp0, mid_publish=False, label0=300, lane0=0, part=False, cont=False, chain=None):
    """mid_publish: behind the test of symbols n/2 - 2 and n/2 - 1 the first n/2 symbols go to the ring and the stream's pixel count
    is published - the first task of the NEXT step (8 pixels: it waits for its left neighbours, the first 8 or 9 symbols of this
    step) starts its late part half a batch earlier (measured: no gain on the chain-bound grids, + 2 ticks per symbol on grid 0).
    part: the block of ONE 8-pixel part of a batch that is decoded part by part (symbols lane0 .. lane0 + 7 of the batch; v50 = the
    part's first row); it ends in the loop's part-end handler (2:) like a part that ran through the 3-copy loop.
    chain (r06, 4-symbol parts): label of the NEXT part's block.  The part asks for the slot's ready word two symbols before its end;
    behind its last symbol it publishes itself and, if the next part's bit is set, goes straight on in that part's block (rows
    4 / 5 of this block are that part's first two: moved to where its block expects them) - ~28 instructions between two parts
    instead of the part-end handler + a look at the ready word + the dispatch (~45 instructions, an LDS round trip, four taken
    branches: ~480 ticks per part measured with 4-pixel tasks on a half-size grid, profiles/r06/prof_grids_t8_40.txt)."""
    assert n % 2 == 0
    out = []
    for j in range(n):
        cur_l, cur_p, nxt = PAIRS[j % 3]
        rcur, rnew = RP[j % 3], RP[(j + 1) % 3]
        dcur, dnew = DP[j % 2], DP[(j + 1) % 2]
        lp_lo, lp_hi, lane = HS[j % 2]
        out.append(q("%d:" % (label0 + j)))
        if part and (cont or chain) and j == n - 2:
            # first part of a batch: ask for the slot's ready word now - behind the block it says whether the second part is there
            out.append(q("ds_read_b32 v54, v51"))
        if mid_publish and j == n // 2:
            out += [q("s_mov_b64 exec, 0x%x" % ((1 << (n // 2)) - 1)), q("v_sub_u32 v52, %[top], %[raw]"), q("v_add_u32 v52, 1, v52"),
                    q("ds_write_b8 %[ring], v52"), q("s_mov_b64 exec, -1"), q("s_add_u32 s58, %[pix0], %[i]"), q("s_add_u32 s58, s58, %d" % (n // 2)),
                    q("v_mov_b32 v56, s58"), q("ds_write_b32 %[rdy], v56 offset:68")]
        wait = "s_waitcnt lgkmcnt(4)" if (mid_publish and j in (n // 2, n // 2 + 1)) else "s_waitcnt lgkmcnt(2)"
        if part and (cont or chain) and j >= n - 2:
            wait = "s_waitcnt lgkmcnt(3)"
        out += [q("s_lshr_b64 s[40:41], %s, 24" % pair(rcur)),
                q("ds_read_b64 %s, v50 offset:%d" % (nxt, 512 * (j + 2))),
                q(wait),
                q("v_mad_u64_u32 v[44:45], s[42:43], s40, %s, 0" % cur_l),
                q("v_mad_u32_u24 v45, %s, s41, v45" % cur_l),
                q("v_cmpx_ge_u64 vcc, %s, v[44:45]" % pair(dcur)),
                q("v_mad_u64_u32 v[48:49], s[42:43], s40, %s, 0" % cur_p),
                q("v_mad_u32_u24 v49, %s, s41, v49" % cur_p),
                q("s_ff1_i32_b64 %s, vcc" % lane),
                q("v_writelane_b32 %%[raw], %s, %d" % (lane, lane0 + j)),
                q("v_readfirstlane_b32 %s, v48" % rnew[0]),
                q("v_readfirstlane_b32 %s, v49" % rnew[1]),
                q("v_readfirstlane_b32 %s, v44" % lp_lo),
                q("v_readfirstlane_b32 %s, v45" % lp_hi),
                q("s_mov_b64 exec, -1")]
        if j % 2:
            out += [q("s_min_u32 s58, %s, %s" % (rcur[1], rnew[1])),  # high words of the new ranges of symbols j - 1 and j
                    q("s_cmp_eq_u32 s58, 0"),
                    q("s_cbranch_scc1 %df" % (tramp0 + j // 2))]
        out += [q("s_sub_u32 %s, %s, %s" % (dnew[0], dcur[0], lp_lo)),
                q("s_subb_u32 %s, %s, %s" % (dnew[1], dcur[1], lp_hi))]
    out.append(q("%d:" % (label0 + n)))
    if part and cont:
        # The first 8 symbols of a 16-pixel batch are decoded: publish them (ring cells + pixel count: the first task of the next step
        # waits for exactly these), then - if the ready word read two symbols ago shows the second part - go on in the SECOND HALF of the
        # full batch's block (its symbol 8 finds rows 8 / 9, the range and the distance where this block leaves them; i still
        # points at the batch's first symbol, as that block expects) and leave through the full batch's epilogue: a batch taken in
        # two parts then costs ~15 instructions more than one taken whole, not a polling round trip and two part hand-overs.
        assert n == 8 and RP[n % 3] == RP[8 % 3] and DP[n % 2] == DP[0]
        out += [q("s_mov_b64 exec, 0xff"), q("v_sub_u32 v52, %[top], %[raw]"), q("v_add_u32 v52, 1, v52"), q("ds_write_b8 %[ring], v52"),
                q("global_store_byte %[goff], v52, %[lat]"), q("s_mov_b64 exec, -1"),
                q("s_add_u32 s58, %[pix0], %[i]"), q("s_add_u32 s58, s58, 8"), q("v_mov_b32 v56, s58"), q("ds_write_b32 %[rdy], v56 offset:68"),
                q("s_sub_u32 s59, s66, %[i]"),
                q("s_waitcnt lgkmcnt(4)"),          # (the ready word: in front of rows 8 and 9 and the two stores)
                q("v_readfirstlane_b32 s58, v54"),
                q("s_cmp_eq_u32 s59, 16"), q("s_cbranch_scc0 %df" % (label0 + n + 1)),
                q("s_bitcmp1_b32 s58, 1"), q("s_cbranch_scc0 %df" % (label0 + n + 1)),
                q("ds_read_b32 %[top], v53"), q("ds_read_b32 %[ring], v53 offset:520"), q("ds_read_b32 %[goff], v53 offset:1040"), q("s_mov_b32 s65, 0"), q("s_mov_b32 s54, s66"), q("s_branch 308b"),
                q("%d:" % (label0 + n + 1)),
                q("s_mov_b64 s[52:53], %s" % pair(RP[n % 3])), q("s_add_u32 %%[i], %%[i], %d" % n), q("s_mov_b32 s67, %[i]"), q("s_branch 126b")]
        return out
    if part and chain:
        assert n == 4 and DP[n % 2] == DP[0] and lane0 in (0, 4, 8)
        nxt_bit = lane0 // 4 + 1
        out += [q("s_mov_b64 exec, 0x%x" % (0xf << lane0)), q("v_sub_u32 v52, %[top], %[raw]"), q("v_add_u32 v52, 1, v52"), q("ds_write_b8 %[ring], v52"),
                q("global_store_byte %[goff], v52, %[lat]"), q("s_mov_b64 exec, -1"),
                q("s_add_u32 s58, %[pix0], %[i]"), q("s_add_u32 s58, s58, 4"), q("v_mov_b32 v56, s58"), q("ds_write_b32 %[rdy], v56 offset:68"),
                q("s_add_u32 s59, %[i], 8"),
                q("s_waitcnt lgkmcnt(4)"),          # (the ready word: in front of rows 4 and 5 and the two stores)
                q("v_readfirstlane_b32 s58, v54"),
                q("s_cmp_le_u32 s59, s66"), q("s_cbranch_scc0 %df" % (label0 + n + 1)),      # a FULL part of this batch behind this one?
                q("s_bitcmp1_b32 s58, %d" % nxt_bit), q("s_cbranch_scc0 %df" % (label0 + n + 1)),
                # go on in the next part's block: range in s[52:53], its rows 0 / 1 in v[40:41] / v[42:43], v50 = its first row, the
                # part's top symbols / ring cells / grid offsets read again (its producer has stored them by now)
                q("s_mov_b64 s[52:53], %s" % pair(RP[n % 3])),
                q("s_waitcnt lgkmcnt(2)"),          # rows 4 and 5
                q("v_mov_b32 v40, v42"), q("v_mov_b32 v41, v43"), q("v_mov_b32 v42, v46"), q("v_mov_b32 v43, v47"),
                q("v_add_u32 v50, 0x800, v50"),
                q("ds_read_b32 %[top], v53"), q("ds_read_b32 %[ring], v53 offset:520"), q("ds_read_b32 %[goff], v53 offset:1040"),
                q("s_add_u32 %[i], %[i], 4"), q("s_mov_b32 s67, %[i]"), q("s_add_u32 s54, %[i], 4"),
                q("s_branch %db" % chain if chain < label0 else "s_branch %df" % chain),
                q("%d:" % (label0 + n + 1)),
                q("s_mov_b64 s[52:53], %s" % pair(RP[n % 3])), q("s_add_u32 %%[i], %%[i], %d" % n), q("s_mov_b32 s67, %[i]"), q("s_branch 126b")]
        return out
    if n % 3:  # the loop and the epilogue expect the range in s[52:53]
        out.append(q("s_mov_b64 s[52:53], %s" % pair(RP[n % 3])))
    out.append(q("s_add_u32 %%[i], %%[i], %d" % n))
    if part:
        out.append(q("s_branch 2b"))
    return out


def loop_conventions(e):
    """Moves for "symbol e left the block": s[50:51] = distance in front of e, s46 / s47 / s44 = its products' low / high word and hit
    lane, s[48:49] = its new range, s[52:53] = the range in front of it; i, v50 as the 3-copy loop wants them; then its handler."""
    out = []
    if DP[e % 2] != DP[0]:
        out.append(q("s_mov_b64 s[50:51], %s" % pair(DP[e % 2])))
    if HS[e % 2] != HS[0]:
        out += [q("s_mov_b32 s46, %s" % HS[1][0]), q("s_mov_b32 s47, %s" % HS[1][1]), q("s_mov_b32 s44, %s" % HS[1][2])]
    before, after = RP[e % 3], RP[(e + 1) % 3]   # -> s[52:53], s[48:49]
    if (before, after) == (RP[0], RP[1]):
        pass
    elif (before, after) == (RP[1], RP[2]):      # s[48:49] -> s[52:53] first, then s[74:75] -> s[48:49]
        out += [q("s_mov_b64 s[52:53], s[48:49]"), q("s_mov_b64 s[48:49], s[74:75]")]
    else:                                         # (RP[2], RP[0]): s[52:53] -> s[48:49] first, then s[74:75] -> s[52:53]
        out += [q("s_mov_b64 s[48:49], s[52:53]"), q("s_mov_b64 s[52:53], s[74:75]")]
    if e:
        out.append(q("s_add_u32 %%[i], %%[i], %d" % e))
    if e // 3:
        out.append(q("v_add_u32 v50, 0x%x, v50" % (0x600 * (e // 3))))
    out.append(q("s_branch %db" % (40 + e % 3)))
    return out


def renorm_in_block(x, odd, lbl_slow, label0=300):
    """Symbol x left with a new range below 2^32.  Not zero (a sentinel goes to the loop's handler): an ordinary renormalisation,
    done here in the block's own registers - commit the symbol, shift the next payload word in - and the block goes on with
    symbol x + 1 (an even x: the odd symbol behind it ran on a useless scale and runs again)."""
    rn = RP[(x + 1) % 3]
    dc, dn = DP[x % 2], DP[(x + 1) % 2]
    lp_lo, lp_hi, _ = HS[x % 2]
    out = [q("s_cmp_eq_u32 %s, 0" % rn[0]),
           q("s_cbranch_scc1 %df" % lbl_slow)]
    if odd:  # the test sits in front of an odd symbol's subtraction
        out += [q("s_sub_u32 %s, %s, %s" % (dn[0], dc[0], lp_lo)), q("s_subb_u32 %s, %s, %s" % (dn[1], dc[1], lp_hi))]
    out += [q("s_sub_u32 s58, %[wpos], %[wbase]"),
            q("s_and_b32 s58, s58, 63"),
            q("v_readlane_b32 s57, %[wbuf], s58"),
            q("s_mov_b32 %s, %s" % (dn[1], dn[0])),
            q("s_mov_b32 %s, s57" % dn[0]),
            q("s_mov_b32 %s, %s" % (rn[1], rn[0])),
            q("s_mov_b32 %s, 0" % rn[0]),
            q("s_add_u32 %[wpos], %[wpos], 1"),
            q("s_cmp_eq_u32 s58, 63"),
            q("s_cbranch_scc0 %db" % (label0 + x + 1))]
    # the 64-word payload buffer is used up: leave for the refill with the loop's conventions, symbol x committed
    if dn != DP[0]:
        out.append(q("s_mov_b64 s[50:51], %s" % pair(dn)))
    if rn != RP[0]:
        out.append(q("s_mov_b64 s[52:53], %s" % pair(rn)))
    out += [q("s_add_u32 %%[i], %%[i], %d" % (x + 1)), q("s_branch 15f")]
    return out


def trampolines_paired(n, tramp0, label0=300):
    out = []
    for k in range(n // 2):
        e, o = 2 * k, 2 * k + 1
        out.append(q("%d:" % (tramp0 + k)))
        out += [q("s_cmp_eq_u32 %s, 0" % RP[(e + 1) % 3][1]),   # the even symbol's new range
                q("s_cbranch_scc0 %df" % (tramp0 + 50 + k))]
        out += renorm_in_block(e, False, tramp0 + 200 + k, label0)
        out.append(q("%d:" % (tramp0 + 200 + k)))
        out += loop_conventions(e)
        out.append(q("%d:" % (tramp0 + 50 + k)))
        out += renorm_in_block(o, True, tramp0 + 250 + k, label0)
        out.append(q("%d:" % (tramp0 + 250 + k)))
        out += loop_conventions(o)
    return out


def main():
    root = Path(__file__).resolve().parents[1] / "cool_chic_amd" / "csrc"
    head = "/* Generated by tools/gen_decoder_block.py - do not edit. */\n"
    for n, tramp0, mid in ((16, 81, False), (32, 101, True)):
        (root / ("ccd_dec_block%d.inc" % n)).write_text(head + "\n".join(block(n, tramp0, mid)) + "\n")
        (root / ("ccd_dec_tramp%d.inc" % n)).write_text(head + "\n".join(trampolines(n, tramp0)) + "\n")
    (root / "ccd_dec_block16p.inc").write_text(head + "\n".join(block_paired(16, 201)) + "\n")
    (root / "ccd_dec_block16pm.inc").write_text(head + "\n".join(block_paired(16, 201, mid_publish=True)) + "\n")
    # the two 8-symbol parts of a 16-pixel batch that is decoded part by part (each with its trampolines behind it)
    parts = []
    for label0, lane0, tramp0 in ((400, 0, 1001), (420, 8, 1401)):
        parts += [q(".p2align 6")] + block_paired(8, tramp0, label0=label0, lane0=lane0, part=True, cont=lane0 == 0) + trampolines_paired(8, tramp0, label0)
    (root / "ccd_dec_parts8.inc").write_text(head + "\n".join(parts) + "\n")
    # the four 4-symbol parts of a 16-pixel batch of 4-pixel tasks (grids whose widest step has 9 .. 24 pixels); r06: chained - and
    # the unchained r05 form beside it for A/B (-DCCD_NO_CHAIN4)
    for chained, fname in ((True, "ccd_dec_parts4.inc"), (False, "ccd_dec_parts4_nc.inc")):
        parts = []
        for k in range(4):
            label0, tramp0 = 440 + 10 * k, 1801 + 400 * k
            parts += [q(".p2align 6")] + block_paired(4, tramp0, label0=label0, lane0=4 * k, part=True, chain=(label0 + 10) if (chained and k < 3) else None) + trampolines_paired(4, tramp0, label0)
        (root / fname).write_text(head + "\n".join(parts) + "\n")
    (root / "ccd_dec_tramp16p.inc").write_text(head + "\n".join(trampolines_paired(16, 201)) + "\n")


if __name__ == "__main__":
    main()
