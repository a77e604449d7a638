#!/usr/bin/env python3
"""Generates pyshepseg_amd/csrc/dfs_walk4_asm.h: the inner loop of the 4-connected replay walker
(clump.h, dfs_split_win<true>) as one GCN assembly block for gfx950.

Why assembly: a walker is ONE wavefront running a dependent chain, so it is paced by instruction
issue -- 4.3-4.5 cycles per scalar or vector instruction for a lone wave, ~10 for a conditional
branch that falls through, ~25 for a taken one, 8.8 for a v_readlane, 52 for an LDS round trip
(tools/ubench/issue.hip) -- and hipcc's structurised control flow spent ~90 instructions and a
dozen branches per step, ~950 cycles between two runs of steps.  Here:

  * a STEP extracts which of the four neighbours are unvisited members (m, 4 bits) from the three
    window rows and branches to one of 15 code blocks; block m knows statically what to clear,
    what to stack, where to move, and ends with the next step's dispatch.  The dispatch is a chain
    of compare-and-branch on m, NOT a jump table: an s_setpc_b64 to another cache line costs ~160
    cycles (the fetch is only started when the address is known), a direct branch 20-25 wherever it
    goes (tools/ubench/branch.hip).  The chain of block p tries the masks in the order they follow
    p on the benchmark imagery (the same mask again in 3 of 4 steps), and skips the bit that is
    known to be clear (the pixel the walk just came from);
  * the GUARD is a step budget: the distance from the position to the rim of the register tile,
    capped by a third of what is left of the 10000-pixel cap (a step marks at most 3 pixels).
    When it runs out the block re-computes it in place (`recomp`: pixels marked = steps + stack
    growth, so the steps carry no counter) and only leaves for the events the C++ side handles:
    tile re-centring, the cap, a full stack window;
  * a DEAD END (m = 0) pops ONE entry here -- window rows back into the tile, the entry's rows out
    of it, no LDS traffic but the 4-byte pop -- and leaves for the bulk test of the stack top (64
    entries at a time, clump.h) only when that entry is dead too or lies outside the tile.

  * RUNS.  The walk repeats itself: on the benchmark imagery 59 % of the steps continue a streak of the SAME mask,
    34 % a streak of four or more (21 %: eight or more), and a tile that is one component is nothing else.  When a
    block's dispatch finds its own mask again (masks 1, 4: straight left / right; 12, 6, 14: right, stacking the
    neighbour below / above / both -- the sweep along the rim of a uniform region; 8, 9, 2, 3: straight down / up,
    stacking the left neighbour or not) it first asks how long the streak is and takes it in one go:
      - left / right: the run is the stretch of set bits of the C row next to the position that has no set bit
        above or below it (the U and D rows): three 64-bit shifts and two bit scans on SGPRs; K bits of C are
        cleared with one s_bfm_b64 / s_andn2_b64;
      - down / up: a row continues the streak iff its three bits around the column read (left as stacked, self 1,
        right 0); with the tile's lane = row that is ONE v_bfe_u32 + v_cmp over all 64 rows, and the streak's
        length a shift and a bit scan of the compare mask.  The K rows are cleared under an exec mask of K lanes,
        and the K stacked left neighbours are written by K lanes with ONE ds_write_b32 (lane i = the i-th step:
        the order the single steps would have pushed them in).
    K is capped by the step budget, so rim, cap and stack room hold as before; a streak shorter than two, or a
    column whose three bits straddle the tile's two dwords, falls through to the plain block.  Equivalence with
    the single steps: the same visiting sequence on 400 random bitmaps in a bit-level model (LABNOTES, round 4),
    and every clump test on the device.

Register contract (see the asm statement in clump.h):
  s[64:65] U, s[66:67] C, s[68:69] D   window rows (64 columns of the tile) above / at / below
  s73 / s74  (in/out) min / max of ryp = tile row + 1 the walk stood on
  s75  (out) why: 0 dead end -> bulk test, 1 re-centre the tile, 2 cap reached, 3 stack window
             nearly full, 4 jumped to an entry outside the tile (window already written back)
  s88  (in/out) pixels marked so far (clumpSize of the reference)
  s89  (in/out) LDS byte address of the stack top
  s90 tr0, s91 twc32 (tile origin: padded row, padded column), s92..s95 slack offsets
       (up = ry - s92, down = s93 - ry, left = b - s94, right = s95 - b; a side on the bitmap's own
       border gets an offset that never binds), s96 end of the stack window, s99 its start
  internal: s70 shw = (b - 1) | 3 << 16, s98 shw1 = b | 1 << 16 (s_bfe_u64 operands), s71 ryp,
       s72 guard, s97 its start value, s[80:85] scratch, s[86:87] exec,
       m0 lane select of v_writelane
  %[tlo] %[thi] the tile (lane = row), %[vcpk] packed position (row << 16 | col), %[vsp] LDS byte
  address of the stack top, %[vt] %[vt2] scratch -- lane 0 only is maintained inside (exec = 1);
  %[vlane] (in) the lane number, all lanes; s76..s79 more scratch (runs)
m bit 0 = left, 1 = up, 2 = right, 3 = down.  Push order of the reference (shepseg.py:519-536, cx
outer, cy inner): left, up, down, right; the walk moves to the last one pushed.
"""
import os

# what follows what (per thousand), measured with the host model of this walk on a 4096 x 4096 tile
# of the benchmark raster (C3); only the ORDER of the tests depends on it
FOLLOW = {
    0: [1, 3, 9, 8, 2], 1: [9, 3, 1, 11, 8], 2: [2, 6, 3, 7, 1], 3: [3, 7, 1, 2, 5], 4: [8, 2, 4, 12, 6],
    5: [8, 2, 4, 12, 6], 6: [2, 6, 14, 10, 4], 7: [2, 6, 4, 8, 12], 8: [8, 9, 12, 13, 1], 9: [9, 1, 13, 8, 5],
    10: [8, 12, 9, 4, 1], 11: [9, 1, 8, 4], 12: [8, 12, 14, 10, 4], 13: [8, 12, 4], 14: [14, 10, 12, 8, 2],
    15: [8, 12, 4],
}
COMMON = [9, 3, 1, 2, 8, 6, 7, 12, 13, 14, 11, 5, 4, 10, 15]


def move_of(m):
    return 'R' if m & 4 else 'Dn' if m & 8 else 'Up' if m & 2 else 'L'


def head(prev):
    """m of the next step into s80 and the branch to its block.  prev: the mask of the block this
    head closes (None: unknown).  After a move the pixel the walk came from is visited, so one bit
    of the next mask is known to be 0 and is not extracted."""
    came = {None: 0, 'R': 1, 'L': 4, 'Dn': 2, 'Up': 8}[move_of(prev) if prev else None]
    o = ['s_bfe_u64 s[80:81], s[66:67], s70']                 # bits 0 and 2: left, right
    if not came & 2:
        o += ['s_bfe_u64 s[82:83], s[64:65], s98', 's_lshl1_add_u32 s80, s82, s80']
    if not came & 8:
        o += ['s_bfe_u64 s[84:85], s[68:69], s98', 's_lshl3_add_u32 s80, s84, s80']
    cand = [m for m in (FOLLOW[prev if prev else 0] + COMMON) if not m & came]
    seen = []
    for m in cand:
        if m not in seen:
            seen.append(m)
    def target(m):
        return '.Ldw4_r%d%%=' % m if (RUNS and m == prev and m in RUNNABLE) else '.Ldw4_b%d%%=' % m
    for m in seen[:-1]:
        o += ['s_cmp_eq_u32 s80, %d' % m, 's_cbranch_scc1 ' + target(m)]
    # the last candidate needs no test unless the mask may be 0 (dead end)
    o += ['s_cmp_eq_u32 s80, %d' % seen[-1], 's_cbranch_scc1 ' + target(seen[-1]), 's_branch .Ldw4_dead%=']
    return o


RUNS = os.environ.get('DFS_WALK_RUNS', '1') != '0'
RUNNABLE = (1, 4, 12, 6, 14, 8, 9, 2, 3)


def hrun(m):
    """a streak of mask 1 (left), 4 (right) or 12 / 6 / 14 (right, stacking the lower / upper / both neighbours):
    K steps along the C row in one go"""
    left = m == 1
    wu, wd = bool(m & 2), bool(m & 8)
    o = ['.Ldw4_r%d%%=:' % m, 's_and_b32 s76, s98, 0xffff']                       # b
    if left:
        o += ['s_sub_u32 s77, 64, s76',
              's_lshl_b64 s[80:81], s[66:67], s77',         # bit b - 1 of C -> bit 63
              's_not_b64 s[80:81], s[80:81]',
              's_flbit_i32_b64 s78, s[80:81]',              # set bits of C from b - 1 downwards
              's_or_b64 s[82:83], s[64:65], s[68:69]',
              's_sub_u32 s77, 63, s76',
              's_lshl_b64 s[82:83], s[82:83], s77',         # bit b of U | D -> bit 63
              's_flbit_i32_b64 s79, s[82:83]']              # clear bits of U | D from b downwards (-1: all)
    else:
        # columns where the rows above / below do NOT read as the mask says end the streak
        bad = {(False, False): 's_or_b64 s[82:83], s[64:65], s[68:69]',
               (False, True): 's_orn2_b64 s[82:83], s[64:65], s[68:69]',          # U | ~D
               (True, False): 's_orn2_b64 s[82:83], s[68:69], s[64:65]',          # D | ~U
               (True, True): 's_nand_b64 s[82:83], s[64:65], s[68:69]'}[(wu, wd)]
        o += ['s_add_u32 s77, s76, 1',
              's_lshr_b64 s[80:81], s[66:67], s77',         # bit b + 1 of C -> bit 0
              's_not_b64 s[80:81], s[80:81]',
              's_ff1_i32_b64 s78, s[80:81]',
              bad,
              's_lshr_b64 s[82:83], s[82:83], s76',
              's_ff1_i32_b64 s79, s[82:83]']
    o += ['s_cmp_lt_i32 s79, 0',
          's_cselect_b32 s79, 64, s79',
          's_min_u32 s78, s78, s79',
          's_min_u32 s78, s78, s72',                        # the step budget
          's_cmp_lt_u32 s78, 2',
          's_cbranch_scc1 .Ldw4_b%d%%=' % m]
    if left:
        o += ['s_sub_u32 s77, s76, s78',                    # lowest bit to clear: b - K
              's_bfm_b64 s[80:81], s78, s77',
              's_andn2_b64 s[66:67], s[66:67], s[80:81]',
              's_sub_u32 s70, s70, s78', 's_sub_u32 s98, s98, s78',
              'v_subrev_u32 %[vcpk], s78, %[vcpk]']
    else:
        o += ['s_bfm_b64 s[80:81], s78, s77',               # K bits from b + 1
              's_andn2_b64 s[66:67], s[66:67], s[80:81]']
        if wu or wd:
            # the K neighbours above / below columns b .. b + K - 1 are marked and stacked: lane i = the i-th step,
            # its entries where the single steps would have put them (up before down)
            nst = int(wu) + int(wd)
            o += ['s_bfm_b64 s[82:83], s78, s76']
            if wu:
                o += ['s_andn2_b64 s[64:65], s[64:65], s[82:83]']
            if wd:
                o += ['s_andn2_b64 s[68:69], s[68:69], s[82:83]']
            o += ['v_readfirstlane_b32 s82, %[vcpk]',
                  'v_readfirstlane_b32 s83, %[vsp]',
                  's_bfm_b64 exec, s78, 0',
                  'v_lshl_add_u32 %%[vt2], %%[vlane], %d, s83' % (2 if nst == 1 else 3)]
            off = 0
            if wu:
                o += ['s_sub_u32 s84, s82, 0x10000',
                      'v_add_u32 %[vt], s84, %[vlane]',
                      'ds_write_b32 %[vt2], %[vt]']
                off = 4
            if wd:
                o += ['s_add_u32 s84, s82, 0x10000',
                      'v_add_u32 %[vt], s84, %[vlane]',
                      'ds_write_b32 %%[vt2], %%[vt] offset:%d' % off]
            o += ['s_mov_b64 exec, 1',
                  's_lshl_b32 s83, s78, %d' % (2 if nst == 1 else 3),
                  'v_add_u32 %[vsp], s83, %[vsp]']
        o += ['s_add_u32 s70, s70, s78', 's_add_u32 s98, s98, s78',
              'v_add_u32 %[vcpk], s78, %[vcpk]']
    o += ['s_sub_u32 s72, s72, s78']
    return o + head(m)


def vrun(m):
    """a streak of mask 8 / 9 (down) or 2 / 3 (up; 9 and 3 stack the left neighbour): K steps in one go"""
    stack_l, down = bool(m & 1), bool(m & 8)
    want = 2 | (1 if stack_l else 0)

    def body(tile, sfx):
        o = ['s_sub_u32 m0, s71, 2',                        # the window goes back into the tile
             'v_writelane_b32 %[tlo], s64, m0', 'v_writelane_b32 %[thi], s65, m0',
             's_sub_u32 m0, s71, 1',
             'v_writelane_b32 %[tlo], s66, m0', 'v_writelane_b32 %[thi], s67, m0',
             's_mov_b32 m0, s71',
             'v_writelane_b32 %[tlo], s68, m0', 'v_writelane_b32 %[thi], s69, m0',
             's_mov_b64 exec, -1',
             'v_bfe_u32 %%[vt], %s, s77, 3' % tile,         # every row's (left, self, right) around the column
             'v_cmp_eq_u32_e64 s[80:81], %%[vt], %d' % want]
        if down:
            o += ['s_lshr_b64 s[80:81], s[80:81], s71',     # row ry + 1 -> bit 0
                  's_not_b64 s[80:81], s[80:81]',
                  's_ff1_i32_b64 s78, s[80:81]']
        else:
            o += ['s_sub_u32 s79, 65, s71',
                  's_lshl_b64 s[80:81], s[80:81], s79',     # row ry - 1 -> bit 63
                  's_not_b64 s[80:81], s[80:81]',
                  's_flbit_i32_b64 s78, s[80:81]']
        o += ['s_min_u32 s78, s78, s72',
              's_cmp_lt_u32 s78, 2',
              's_cbranch_scc1 .Ldw4_r%dno%s%%=' % (m, sfx),
              's_lshl_b32 s84, 1, s77',                     # the left bit inside the dword, the self bit
              's_lshl_b32 s85, s84, 1',
              's_sub_u32 s79, s71, 1']                      # ry
        if down:
            if stack_l:
                o += ['s_bfm_b64 exec, s78, s79', 's_not_b32 s82, s84', 'v_and_b32 %s, s82, %s' % (tile, tile)]
            o += ['s_bfm_b64 exec, s78, s71', 's_not_b32 s82, s85', 'v_and_b32 %s, s82, %s' % (tile, tile)]
        else:
            o += ['s_sub_u32 s83, s79, s78']                # ry - K
            if stack_l:
                o += ['s_add_u32 s82, s83, 1', 's_bfm_b64 exec, s78, s82', 's_not_b32 s82, s84',
                      'v_and_b32 %s, s82, %s' % (tile, tile)]
            o += ['s_bfm_b64 exec, s78, s83', 's_not_b32 s82, s85', 'v_and_b32 %s, s82, %s' % (tile, tile)]
        o += ['s_mov_b64 exec, 1']
        if stack_l:
            o += ['v_readfirstlane_b32 s82, %[vcpk]',
                  'v_readfirstlane_b32 s83, %[vsp]',
                  's_sub_u32 s82, s82, 1',                  # the left neighbour of the first step
                  's_bfm_b64 exec, s78, 0']                 # lane i = step i
            if down:
                o += ['v_lshl_add_u32 %[vt], %[vlane], 16, s82']
            else:
                o += ['v_lshlrev_b32 %[vt], 16, %[vlane]', 'v_sub_u32 %[vt], s82, %[vt]']
            o += ['v_lshl_add_u32 %[vt2], %[vlane], 2, s83',
                  'ds_write_b32 %[vt2], %[vt]',
                  's_mov_b64 exec, 1',
                  's_lshl_b32 s83, s78, 2',
                  'v_add_u32 %[vsp], s83, %[vsp]']
        o += ['s_lshl_b32 s82, s78, 16']
        if down:
            o += ['v_add_u32 %[vcpk], s82, %[vcpk]', 's_add_u32 s71, s71, s78', 's_max_u32 s74, s74, s71']
        else:
            o += ['v_subrev_u32 %[vcpk], s82, %[vcpk]', 's_sub_u32 s71, s71, s78', 's_min_u32 s73, s73, s71']
        o += ['s_sub_u32 s72, s72, s78',
              's_sub_u32 s80, s71, 2', 's_sub_u32 s81, s71, 1',      # the window at the new row
              'v_readlane_b32 s64, %[tlo], s80', 'v_readlane_b32 s65, %[thi], s80',
              'v_readlane_b32 s66, %[tlo], s81', 'v_readlane_b32 s67, %[thi], s81',
              'v_readlane_b32 s68, %[tlo], s71', 'v_readlane_b32 s69, %[thi], s71']
        o += head(m)
        o += ['.Ldw4_r%dno%s%%=:' % (m, sfx), 's_mov_b64 exec, 1', 's_branch .Ldw4_b%d%%=' % m]
        return o

    o = ['.Ldw4_r%d%%=:' % m,
         's_and_b32 s76, s98, 0xffff',                      # b
         's_cmp_gt_u32 s76, 30',
         's_cbranch_scc1 .Ldw4_r%dhi%%=' % m,
         's_sub_u32 s77, s76, 1']                           # bits b - 1 .. b + 1 in the low dword
    o += body('%[tlo]', 'a')
    o += ['.Ldw4_r%dhi%%=:' % m,
          's_cmp_lt_u32 s76, 33',
          's_cbranch_scc1 .Ldw4_b%d%%=' % m,                # the three bits straddle the dwords: a plain step
          's_sub_u32 s77, s76, 33']
    o += body('%[thi]', 'b')
    return o


MBITS = [
    's_bfe_u64 s[80:81], s[66:67], s70',
    's_bfe_u64 s[82:83], s[64:65], s98',
    's_bfe_u64 s[84:85], s[68:69], s98',
    's_lshl1_add_u32 s80, s82, s80',
    's_lshl3_add_u32 s80, s84, s80',
]
SETTLE = [                                  # pixels marked since the last settle = steps + pushes
    's_sub_u32 s80, s97, s72',
    'v_readfirstlane_b32 s81, %[vsp]',
    's_add_u32 s88, s88, s80',
    's_sub_u32 s80, s81, s89',
    's_lshr_b32 s80, s80, 2',
    's_add_u32 s88, s88, s80',
    's_mov_b32 s89, s81',
]
RECOMP_ENTRY = [
    's_cmp_ge_u32 s88, 0x2710',
    's_cbranch_scc1 .Ldw4_xcap%=',
    's_sub_u32 s80, s96, s89',
    's_cmp_lt_u32 s80, 512',                # room for what the budget allows: 61 steps of two entries, and the seed's third
    's_cbranch_scc1 .Ldw4_xspill%=',
    'v_readfirstlane_b32 s81, %[vcpk]',
    's_lshr_b32 s82, s81, 16',
    's_and_b32 s83, s81, 0xffff',
    's_sub_u32 s82, s82, s90',              # ry
    's_sub_u32 s83, s83, s91',              # b
    's_sub_i32 s80, s82, s92',
    's_sub_i32 s81, s93, s82',
    's_min_i32 s80, s80, s81',
    's_sub_i32 s81, s83, s94',
    's_min_i32 s80, s80, s81',
    's_sub_i32 s81, s95, s83',
    's_min_i32 s80, s80, s81',              # steps to the rim
    's_cmp_lt_i32 s80, 6',
    's_cbranch_scc1 .Ldw4_xretile%=',
    's_sub_u32 s81, 0x270f, s88',
    's_mul_hi_u32 s81, s81, 0xaaaaaaab',
    's_lshr_b32 s81, s81, 1',               # (9999 - marked) / 3
    's_min_u32 s97, s80, s81',
    's_mov_b32 s72, s97',
    's_add_u32 s71, s82, 1',
    's_sub_u32 s70, s83, 1',
    's_or_b32 s70, s70, 0x30000',
    's_or_b32 s98, s83, 0x10000',
]
DEAD = SETTLE + [
    's_cmp_eq_u32 s89, s99',
    's_cbranch_scc1 .Ldw4_xdead%=',         # the LDS window is empty: refill / end of the piece
    's_sub_u32 s89, s89, 4',
    'v_mov_b32 %[vsp], s89',
    'ds_read_b32 %[vt], %[vsp]',
    's_sub_u32 m0, s71, 2',                 # meanwhile: the window goes back into the tile
    'v_writelane_b32 %[tlo], s64, m0',
    'v_writelane_b32 %[thi], s65, m0',
    's_sub_u32 m0, s71, 1',
    'v_writelane_b32 %[tlo], s66, m0',
    'v_writelane_b32 %[thi], s67, m0',
    's_mov_b32 m0, s71',
    'v_writelane_b32 %[tlo], s68, m0',
    'v_writelane_b32 %[thi], s69, m0',
    's_waitcnt lgkmcnt(0)',
    'v_readfirstlane_b32 s80, %[vt]',
    'v_mov_b32 %[vcpk], %[vt]',
    's_lshr_b32 s81, s80, 16',
    's_and_b32 s82, s80, 0xffff',
    's_sub_u32 s81, s81, s90',              # ry
    's_sub_u32 s82, s82, s91',              # b
    's_sub_u32 s83, s81, 1',
    's_cmp_ge_u32 s83, 62',
    's_cbranch_scc1 .Ldw4_xjump%=',
    's_sub_u32 s84, s82, 1',
    's_cmp_ge_u32 s84, 62',
    's_cbranch_scc1 .Ldw4_xjump%=',
    's_add_u32 s71, s81, 1',
    's_min_u32 s73, s73, s71',
    's_max_u32 s74, s74, s71',
    'v_readlane_b32 s64, %[tlo], s83',
    'v_readlane_b32 s65, %[thi], s83',
    'v_readlane_b32 s66, %[tlo], s81',
    'v_readlane_b32 s67, %[thi], s81',
    'v_readlane_b32 s68, %[tlo], s71',
    'v_readlane_b32 s69, %[thi], s71',
    's_or_b32 s70, s84, 0x30000',
    's_or_b32 s98, s82, 0x10000',
] + MBITS + [
    's_cmp_eq_u32 s80, 0',
    's_cbranch_scc1 .Ldw4_xdead%=',         # dead as well: the bulk test is cheaper from here
    's_branch .Ldw4_recomp_entry%=',
]


def block(m):
    L, Up, R, Dn = m & 1, m & 2, m & 4, m & 8
    move = 'R' if R else 'Dn' if Dn else 'Up' if Up else 'L'
    order = [('L', L), ('Up', Up), ('Dn', Dn), ('R', R)]
    stacked = [n for n, a in order if a and n != move]
    o = []
    cmask = (1 if L else 0) | (4 if R else 0)
    if cmask:
        o += ['s_lshl_b64 s[80:81], %d, s70' % cmask, 's_andn2_b64 s[66:67], s[66:67], s[80:81]']
    if Up or Dn:
        o += ['s_lshl_b64 s[82:83], 1, s98']
        if Up:
            o += ['s_andn2_b64 s[64:65], s[64:65], s[82:83]']
        if Dn:
            o += ['s_andn2_b64 s[68:69], s[68:69], s[82:83]']
    delta = {'L': '-1', 'Up': '0xffff0000', 'Dn': '0x10000'}
    for j, n in enumerate(stacked):
        o += ['v_add_u32 %%[vt], %s, %%[vcpk]' % delta[n], 'ds_write_b32 %%[vsp], %%[vt] offset:%d' % (4 * j)]
    if stacked:
        o += ['v_add_u32 %%[vsp], %d, %%[vsp]' % (4 * len(stacked))]
    if move == 'R':
        o += ['s_add_u32 s70, s70, 1', 's_add_u32 s98, s98, 1', 'v_add_u32 %[vcpk], 1, %[vcpk]']
    elif move == 'L':
        o += ['s_sub_u32 s70, s70, 1', 's_sub_u32 s98, s98, 1', 'v_add_u32 %[vcpk], -1, %[vcpk]']
    elif move == 'Dn':
        o += ['s_sub_u32 m0, s71, 2',                      # row leaving the window: the old U
              'v_writelane_b32 %[tlo], s64, m0',
              'v_writelane_b32 %[thi], s65, m0',
              's_mov_b64 s[64:65], s[66:67]',
              's_mov_b64 s[66:67], s[68:69]',
              's_add_u32 s71, s71, 1',
              'v_readlane_b32 s68, %[tlo], s71',
              'v_readlane_b32 s69, %[thi], s71',
              's_max_u32 s74, s74, s71',
              'v_add_u32 %[vcpk], 0x10000, %[vcpk]']
    else:
        o += ['s_mov_b32 m0, s71',                         # row leaving the window: the old D
              'v_writelane_b32 %[tlo], s68, m0',
              'v_writelane_b32 %[thi], s69, m0',
              's_mov_b64 s[68:69], s[66:67]',
              's_mov_b64 s[66:67], s[64:65]',
              's_sub_u32 s71, s71, 1',
              's_sub_u32 s80, s71, 2',
              'v_readlane_b32 s64, %[tlo], s80',
              'v_readlane_b32 s65, %[thi], s80',
              's_min_u32 s73, s73, s71',
              'v_add_u32 %[vcpk], 0xffff0000, %[vcpk]']
    o += ['s_sub_u32 s72, s72, 1', 's_cbranch_scc1 .Ldw4_recomp%=']
    return ['.Ldw4_b%d%%=:' % m] + o + head(m)


def main():
    # (only lane 0 works inside: one stack slot, one position)
    lines = ['s_mov_b64 s[86:87], exec', 's_mov_b64 exec, 1', 's_branch .Ldw4_recomp_entry%=',
             '.Ldw4_recomp%=:'] + SETTLE + ['.Ldw4_recomp_entry%=:'] + RECOMP_ENTRY + head(None)
    lines += ['.Ldw4_dead%=:'] + DEAD
    for i, name in enumerate(['xdead', 'xretile', 'xcap', 'xspill', 'xjump']):
        lines += ['.Ldw4_%s%%=:' % name, 's_mov_b32 s75, %d' % i, 's_branch .Ldw4_exit%=']
    for m in range(1, 16):
        lines += block(m)
    if RUNS:
        for m in (1, 4, 12, 6, 14):
            lines += hrun(m)
        for m in (8, 9, 2, 3):
            lines += vrun(m)
    lines += ['.Ldw4_exit%=:', 's_mov_b64 exec, s[86:87]']
    out = ['// GENERATED by tools/gen_dfs_walk4.py -- do not edit; see that script for the design.',
           '#pragma once', '#define DFS_WALK4_ASM \\']
    for ln in lines:
        out.append('    "%s\\n" \\' % ln)
    out.append('    ""')
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pyshepseg_amd', 'csrc', 'dfs_walk4_asm.h')
    path = os.environ.get('DFS_WALK_OUT', path)             # (tests/test_abi.py: the committed header is this script's output)
    with open(path, 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote', path, len(lines), 'lines')


if __name__ == '__main__':
    main()
