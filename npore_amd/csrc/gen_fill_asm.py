#!/usr/bin/env python3
"""Generator of fill_step_asm.inc: the plain ("FAST") anti-diagonal loop of fill_kernel as hand-scheduled gfx950
assembly, one text per wave role (0 = only wave of a chunk, 1 = first, 2 = middle, 3 = last of several).

Why assembly: the compiler places 15 ... 31 register copies per step where the bodies of the two step kinds ('I' / 'D'
of the input path) meet -- a fifth of the step's vector instructions -- and no source-level
formulation removed them.  Here every loop-carried value has ONE register for the whole loop (the asm operands), both
step kinds update it in place, and the temporaries are fixed scratch registers (v91 ... v111, declared clobbered).

What the text does is exactly the `step` lambda of kernels.hpp for FASTSEL = true (cell.hpp cell_update<FAST>), for as
many steps of a span as it can take, in BLOCKS (to the end of the 64-step window, of the span, or of what the word queues
and the L window hold: `block_end`): it returns with status 1 in front of a step that needs the rare path (a column
descriptor with DSC_RARE: tested once per entry and, in a 'D' step, on the descriptor that enters) -- nothing of that step
is done yet -- or with status 2 behind the poll of a 'D' step whose entering descriptor (learnt from the wave above) is
rare, and the caller runs that one step through the C++ body and comes back.
Several waves per chunk: the hand-over writes only the exchange record the NEXT step reads (`finish`), the first look at
the neighbours' progress words and the read of the neighbour's boundary cell are issued at the head of the step
(`poll_issue`), a wave that has to wait polls at issue priority 0 (`polls`).

    python npore_amd/csrc/gen_fill_asm.py        # rewrites fill_step_asm.inc next to this file

Hazards of gfx940-class hardware that the assembler does not resolve for inline text are handled by `fix_hazards`
(VALU-written SGPR / VCC read by a VALU within 2 issue slots, a VGPR written by a VALU read by a DPP move within 2 or by
v_readlane within 1): it inserts s_nop where the written schedule does not already keep the distance.
"""
import os
import re

# ---- scratch registers (clobbered) ---------------------------------------------------------------------------------
# (X0 ... X3 are an aligned quad, X4 X5 an aligned pair: the exchange words of the neighbour waves arrive by ds_read_b128 / ds_read2_b32)
X0, X1, X2, X3, X4, X5 = "v92", "v93", "v94", "v95", "v96", "v97"      # neighbour cell / exchange words, then scratch
XQ, X01, X23, X45 = "v[92:95]", "v[92:93]", "v[94:95]", "v[96:97]"
SD, SE, SF = "v91", "v98", "v99"
P0, P1, PP = "v100", "v101", "v[100:101]"
SUBV, DRUN, NINSR, NDELR = "v102", "v103", "v87", "v88"
SHRV, SHRRUN, LENV, LENRUN = "v104", "v105", "v106", "v107"
Q0, LENST, SHRST, QRUNS, QQ = "v108", "v109", "v110", "v111", "v[108:111]"      # the history record: one ds_write_b128
# MAT.VAL of the lane's cell LIVES in the record's first register for the whole loop (copied in from / out to the operand
# `matv` where the text is entered / left): the step's last 3-way minimum writes it there and the record is stored from
# there -- no copy per step.  Like LENST it is a data register of the ds_write_b128: written again only behind a wait.
MATV = Q0
HS, HP0, HP1, HQ = "v104", "v106", "v107", "v[104:107]"      # the first SHR candidate's source record (one ds_read_b128): matv, -, shrstart, runs
SMR = SF      # the descriptor's summary bits (rc0 & 0xbc) from the head of a step to the SHR pass: NOT one of the record's
              # registers v108 ... v111, which the previous step's ds_write_b128 may still be reading there
SCRATCH = ["v%d" % k for k in range(87, 112)]

LDS_SUB_BASE = 6 * 32 * 33 * 4      # kernels.hpp LDS_SUB_BASE
XCH_WORDS = 12                      # kernels.hpp


def O(name):
    return "%[" + name + "]"


class Text:
    """main = the path most steps take (falls through from label to label); ool = rare blocks, emitted behind it"""
    def __init__(self):
        self.lines = []
        self.main = self.lines
        self.ool = []

    def rare(self):
        self.lines = self.ool

    def common(self):
        self.lines = self.main

    def __call__(self, s):
        for ln in s.strip("\n").split("\n"):
            ln = ln.strip()
            if ln:
                self.lines.append(ln)

    def label(self, name):
        self.lines.append(name + "_%=:")


def L(name):
    return name + "_%="


# more scratch: lane-table results of the first SHR candidate, fetched before the hand-shake poll
E0, E1 = "v89", "v90"
ABLATE = {"nopoll": False, "nolen": False}      # timing-only ablations (wrong strings): --nopoll / --nolen with --out FILE


def shr_tables(t, tmp=X3):
    """lane tables of the column's first SHR candidate (cell.hpp shr_small): where its source record lies, 1/n.
    tmp: a register with NO load on its way (a VALU write to a register an LDS read is still going to fill races)"""
    t(f"""
        v_and_b32 {tmp}, 28, {O('rc0')}
        ds_bpermute_b32 {E0}, {tmp}, {O('tab')}
        ds_bpermute_b32 {E1}, {tmp}, {O('trecip')}
    """)


def shr_hist(t):
    t(f"""
        v_add_u32 {E0}, {O('hca')}, {E0}
        ds_read_b128 {HQ}, {E0}
    """)


def sub_read(t):
    t(f"""
        v_alignbit_b32 {SUBV}, {O('refx')}, {O('seqw')}, 25
        v_and_b32 {SUBV}, 0x3fc, {SUBV}
        ds_read_b32 {SUBV}, {SUBV} offset:{LDS_SUB_BASE}
    """)


def two_adds(t, A, B, m, x):
    """A = indel_start + m, B = indel_extend + x (src/aln.pyx:530-531, 552-553).  (One v_pk_add_f32 on aligned register
    pairs was measured: no gain -- LABNOTES round 4.)"""
    t(f"""
        v_add_f32 {A}, {O('istart')}, {m}
        v_add_f32 {B}, {O('iext')}, {x}
    """)


def ins_part(t, mode, A, B, msk, fill=()):
    """INS (src/aln.pyx:525-543): new value straight into the own register, run into NINSR.  A, B: free registers;
    fill: up to two independent instructions for the slots between the compare and its selects"""
    if mode == "I":
        topM, topI, topR = MATV, O("insv"), O("R1")
    else:
        topM, topI, topR = X0, X1, O("TMr")
    two_adds(t, A, B, topM, topI)
    t(f"""
        v_cmp_lt_f32 {msk}, {B}, {A}
        v_add_u32_sdwa {NINSR}, {topR}, {O('oneI')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD
    """)
    for f in fill:
        t(f)
    t(f"""
        v_cndmask_b32 {O('insv')}, {A}, {B}, {msk}
        v_cndmask_b32 {NINSR}, {O('oneI')}, {NINSR}, {msk}
    """)


def del_part(t, mode, A, B, msk):
    """DEL (src/aln.pyx:547-565); leaves X3 = the six LEN pre-filter bits (refx & seqw flags) as its fillers"""
    if mode == "I":
        leftM, leftD, leftR = X0, X1, O("LMr")
    else:
        leftM, leftD, leftR = MATV, O("delv"), O("R2")
    two_adds(t, A, B, leftM, leftD)
    t(f"""
        v_cmp_lt_f32 {msk}, {B}, {A}
        v_add_u32_sdwa {NDELR}, {leftR}, {O('oneD')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD
        v_and_b32 {X3}, {O('refx')}, {O('seqw')}
        v_bfe_u32 {X3}, {X3}, 8, 6
        v_cndmask_b32 {O('delv')}, {A}, {B}, {msk}
        v_cndmask_b32 {NDELR}, {O('oneD')}, {NDELR}, {msk}
    """)


def shr_pass(t, mid, sfx, smr, shadow, shadow2, none_test=True):
    """SHR candidates of the column (cell.hpp shr_small<FAST>).  On entry: the first candidate's lane-table results in
    E0 (address of its source record) / E1 (1/n) and the record itself in SD (matv), P0 (shrstart), P1 (runs).
    shadow(): work issued in the shadow of the score read (free registers X4 X5 SF); shadow2(): the same for the
    two-candidate block (free registers SHRV SHRRUN).  Leaves X3 = refx & seqw (shadow's last act) for the LEN test.
    The single-candidate case falls through; "no candidate in the wave" and "two candidates" are out of line."""
    # "no candidate in the wave": worth a test only where many lanes are dead -- the first and last wave of a wide band
    # (r=100: 6 % of the wave-steps); a lone wave (r <= 31) has one in 99.8 % of its steps, and a lane without one
    # holds the empty descriptor, whose score is +infinity
    if not mid and none_test:
        t(f"""
            v_cmp_ne_u32 vcc, 0, {smr}
            s_cbranch_vccz {L('shr_none' + sfx)}
        """)
        t.rare()
        t.label("shr_none" + sfx)
        t(f"""
            v_mov_b32 {SHRV}, {O('ev')}
            v_mov_b32 {SHRRUN}, {O('tagS')}
            v_mov_b32 {SHRST}, 0x7f800000
        """)
        shadow()
        t(f"s_branch {L('shr_done2' + sfx)}")
        t.common()
    t(f"""
        v_cmp_lt_u32 vcc, 28, {smr}
        s_cbranch_vccnz {L('shr_two' + sfx)}
    """)
    # ---- one candidate per column
    t(f"""
        v_cmp_gt_i32 vcc, 0, {O('rc0')}
        v_lshrrev_b32 {HP1}, 16, {HP1}
        v_bfe_u32 {SE}, {O('rc0')}, 15, 16
        v_cndmask_b32 {HP1}, {HP1}, 0, vcc
        v_cndmask_b32 {HS}, {HP0}, {HS}, vcc
        v_mul_u32_u24 {E1}, {HP1}, {E1}
        v_bfe_u32 {E0}, {O('rc0')}, 2, 3
        v_min_u32_sdwa {E1}, {E1}, {O('rc0')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_1
        v_add3_u32 {HP1}, {HP1}, {E0}, {O('tagS')}
        v_lshl_add_u32 {SE}, {E1}, 2, {SE}
        ds_read_b32 {SE}, {SE}
    """)
    shadow()
    t(f"""
        s_waitcnt lgkmcnt(0)
        v_add_f32 {SE}, {HS}, {SE}
        v_cmp_lt_f32 vcc, {SE}, {O('ev')}
        v_cndmask_b32 {SHRST}, {O('inf')}, {HS}, vcc
        v_cndmask_b32 {SHRRUN}, {O('tagS')}, {HP1}, vcc
        v_cndmask_b32 {SHRV}, {O('ev')}, {SE}, vcc
    """)
    t.label("shr_done2" + sfx)
    # ---- two candidates (second in rc1): its lane tables and record now, then both scores, then the compares in
    # the reference's order
    t.rare()
    t.label("shr_two" + sfx)
    t(f"""
        v_and_b32 {X3}, 28, {O('rc1')}
        ds_bpermute_b32 {X4}, {X3}, {O('tab')}
        ds_bpermute_b32 {X5}, {X3}, {O('trecip')}
        v_cmp_gt_i32 vcc, 0, {O('rc0')}
        v_cmp_gt_i32 {O('sa')}, 0, {O('rc1')}
        v_lshrrev_b32 {HP1}, 16, {HP1}
        v_bfe_u32 {SE}, {O('rc0')}, 15, 16
        v_cndmask_b32 {HP1}, {HP1}, 0, vcc
        v_cndmask_b32 {HS}, {HP0}, {HS}, vcc
        v_mul_u32_u24 {E1}, {HP1}, {E1}
        v_bfe_u32 {E0}, {O('rc0')}, 2, 3
        v_min_u32_sdwa {E1}, {E1}, {O('rc0')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_1
        v_add3_u32 {HP1}, {HP1}, {E0}, {O('tagS')}
        v_lshl_add_u32 {E0}, {E1}, 2, {SE}
        s_waitcnt lgkmcnt(0)
        v_add_u32 {X4}, {O('hca')}, {X4}
        ds_read_b32 {E1}, {X4}
        ds_read_b64 v[98:99], {X4} offset:8
        ds_read_b32 {E0}, {E0}
    """)
    shadow2()                # (X4 X5 SE SF E0 E1 SD P0 P1 are in use; leaves X3 = refx & seqw)
    t(f"""
        s_waitcnt lgkmcnt(0)
        v_lshrrev_b32 {SF}, 16, {SF}
        v_cndmask_b32 {SF}, {SF}, 0, {O('sa')}
        v_cndmask_b32 {E1}, {SE}, {E1}, {O('sa')}
        v_mul_u32_u24 {X5}, {SF}, {X5}
        v_bfe_u32 {X4}, {O('rc1')}, 15, 16
        v_min_u32_sdwa {X5}, {X5}, {O('rc1')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_1
        v_bfe_u32 {SE}, {O('rc1')}, 2, 3
        v_lshl_add_u32 {X4}, {X5}, 2, {X4}
        ds_read_b32 {X4}, {X4}
        v_add3_u32 {SF}, {SF}, {SE}, {O('tagS')}
        v_add_f32 {E0}, {HS}, {E0}
        v_cmp_lt_f32 vcc, {E0}, {O('ev')}
        v_cndmask_b32 {SHRST}, {O('inf')}, {HS}, vcc
        v_cndmask_b32 {SHRRUN}, {O('tagS')}, {HP1}, vcc
        v_cndmask_b32 {SHRV}, {O('ev')}, {E0}, vcc
        s_waitcnt lgkmcnt(0)
        v_add_f32 {X4}, {E1}, {X4}
        v_cmp_lt_f32 vcc, {X4}, {SHRV}
        s_nop 1
        v_cndmask_b32 {SHRV}, {SHRV}, {X4}, vcc
        v_cndmask_b32 {SHRRUN}, {SHRRUN}, {SF}, vcc
        v_cndmask_b32 {SHRST}, {SHRST}, {E1}, vcc
        s_branch {L('shr_done2' + sfx)}
    """)
    t.common()


def len_pass(t, mid, sfx, mode, first, last, multi):
    """LEN candidates (cell.hpp, LEN loop; LEN_ARITH form).  On entry X3 = the six "read position i-n in an n-polymer
    and reference position j starts one" bits.  The loop is out of line (two steps in three have no candidate).
    Free: X4 X5 SD SE SF P0 P1 E0 E1.  (Lengths of 32 and more -- `big` -- are read from the full table in global memory; the
    LDS read of that lane then lands somewhere in the LDS and is not used: its index is formed without range masks.)"""
    if not mid:
        t(f"v_cndmask_b32 {X3}, 0, {X3}, {O('mhist')}")
    if ABLATE["nolen"]:
        t(f"v_mov_b32 {X3}, 0")
    t(f"""
        v_cmp_ne_u32 vcc, 0, {X3}
        s_cbranch_vccnz {L('len_body' + sfx)}
    """)
    t.rare()
    t.label("len_top" + sfx)
    t(f"""
        v_cmp_ne_u32 vcc, 0, {X3}
        s_cbranch_vccz {L('len_done' + sfx)}
        s_branch {L('len_iter' + sfx)}
    """)
    t.label("len_body" + sfx)
    t(f"""
        v_mov_b32 {LENV}, {O('ev')}
        v_mov_b32 {LENRUN}, 0
    """)
    t.label("len_iter" + sfx)
    t(f"""
        s_mov_b64 {O('sa')}, vcc
        v_ffbh_u32 {X4}, {X3}
        v_sub_u32 {X4}, 31, {X4}
        v_bfe_u32 {X3}, {X3}, 0, {X4}
        v_mad_i32_i24 {X5}, {X4}, -3, 29
        v_lshrrev_b32 {SD}, 14, {O('refx')}
        v_lshrrev_b32 {SE}, {X5}, {O('seqw')}
        v_xor_b32 {SE}, {SE}, {SD}
        v_lshlrev_b32 {SE}, {X5}, {SE}
        v_cmp_eq_u32 {O('sb')}, 0, {SE}
        s_and_b64 {O('sb')}, {O('sb')}, {O('sa')}
        s_cbranch_scc0 {L('len_top' + sfx)}
        v_add_u32 {SF}, 1, {X4}
        v_lshlrev_b32 {E0}, 2, {SF}
        ds_bpermute_b32 {E1}, {E0}, {O('tab')}
        ds_bpermute_b32 {X5}, {E0}, {O('trecip')}
        v_bfe_u32 {SD}, {O('seqw')}, {X4}, 1
        v_cmp_ne_u32 {O('sc')}, 0, {SD}
        v_add_u32 {SD}, {O('sdel')}, {O('lanej')}
        v_and_b32 {SD}, {O('wmask')}, {SD}
        v_and_b32 {X4}, 7, {X4}
        v_lshl_add_u32 {SD}, {SD}, 3, {X4}
        v_add_u32 {SD}, {O('winaddr')}, {SD}
        ds_read_u8 {SD}, {SD}
        s_waitcnt lgkmcnt(0)
        v_lshl_add_u32 {E1}, {E0}, 2, {E1}
        v_add_u32 {E1}, {O('hca')}, {E1}
        ds_read_b64 {PP}, {E1}
        ds_read_b32 {E0}, {E1} offset:12
        v_cmp_ne_u32 {O('sa')}, 0, {SD}
        v_cmp_ge_u32 vcc, {O('clamp1')}, {SF}
        s_and_b64 {O('sa')}, {O('sa')}, vcc
        s_waitcnt lgkmcnt(0)
        v_cndmask_b32 {P0}, {P1}, {P0}, {O('sc')}
        v_and_b32 {E0}, 0xffff, {E0}
        v_cndmask_b32 {E0}, {E0}, 0, {O('sc')}
        v_mul_u32_u24 {P1}, {E0}, {X5}
        v_lshrrev_b32 {P1}, 16, {P1}
        v_add3_u32 {P1}, {SD}, {P1}, 1
        v_min_u32 {SD}, {O('clampv')}, {SD}
        v_min_u32 {P1}, {O('clampv')}, {P1}
        v_or_b32 {SE}, {SD}, {P1}
        v_cmp_lt_u32 vcc, 31, {SE}
        v_lshl_or_b32 {SE}, {X4}, 5, {SD}
        v_mul_u32_u24 {SE}, 33, {SE}
        v_sub_u32 {SE}, {SE}, {P1}
        v_add_u32 {SE}, 31, {SE}
        v_lshlrev_b32 {SE}, 2, {SE}
        ds_read_b32 {SE}, {SE}
        s_and_b64 vcc, vcc, {O('sb')}
        s_cbranch_vccz {L('len_lds' + sfx)}
        s_waitcnt lgkmcnt(0)
        s_mov_b64 exec, vcc
        v_mul_lo_u32 {X5}, {X4}, {O('npdim')}
        v_add_u32 {X5}, {X5}, {SD}
        v_mul_lo_u32 {X5}, {X5}, {O('npdim')}
        v_add_u32 {X5}, {X5}, {P1}
        v_lshlrev_b32 {X5}, 2, {X5}
        global_load_dword {SE}, {X5}, {O('gnp')}
        s_waitcnt vmcnt(0)
        s_mov_b64 exec, -1
    """)
    t.label("len_lds" + sfx)
    t(f"""
        s_waitcnt lgkmcnt(0)
        s_mov_b64 vcc, {O('sa')}
        v_cndmask_b32 {SE}, {O('c100')}, {SE}, vcc
        v_add_f32 {SE}, {P0}, {SE}
        v_add_u32 {E0}, {E0}, {SF}
        v_cmp_lt_f32 vcc, {SE}, {LENV}
        s_and_b64 vcc, vcc, {O('sb')}
        v_cndmask_b32 {LENV}, {LENV}, {SE}, vcc
        v_cndmask_b32 {LENRUN}, {LENRUN}, {E0}, vcc
        v_cndmask_b32 {LENST}, {LENST}, {P0}, vcc
        s_branch {L('len_top' + sfx)}
    """)
    t.label("len_done" + sfx)
    mat_part(t, mode, True, mid, first, last)
    # The record's ds_write_b128 has just been issued, and LENST is one of its four data registers: an LDS write of
    # more than 64 bits reads its data registers over several cycles after issue, like the VMEM stores of the ISA
    # manual's hazard table (which does not list DS) -- a VALU write right behind it can reach the register first (seen
    # as records whose LEN run start was +inf in some launches; LABNOTES round 3).  The architected guarantee is the
    # counter: LENST is restored only behind an lgkmcnt wait that retires the write (the LDS unit serves a wave's
    # requests in order).  Several waves per chunk: this variant carries its own copy of the hand-over and the
    # progress store, and the wait in front of the progress word is that wait; a lone wave waits here (its three neighbours on the SIMD fill the gap).
    if multi:
        finish(t, mode, first, last, True, mid, True)      # (LENST is restored where all lanes are enabled again)
    else:
        t(f"""
            s_waitcnt lgkmcnt(0)
            v_mov_b32 {LENST}, 0x7f800000
            s_branch {L('post_mat_' + mode)}
        """)
    t.common()


def mat_part(t, mode, with_len, mid, first=False, last=False):
    """MAT by two 3-way minima and equality tests (cell.hpp, Env::MIN3): value, traceback word, runs.  with_len = False:
    no LEN candidate in the wave -- LEN.VAL is 100 b, its run 0, its run start +inf (LENST keeps that value from the
    loop's entry on); True: the out-of-line variant behind the LEN loop, which reads LENV / LENRUN and restores LENST."""
    own_m, own_i, own_d, r1, r2 = MATV, O("insv"), O("delv"), O("R1"), O("R2")
    diagM = O("LMv") if mode == "I" else O("TMv")
    lenv = LENV if with_len else O("ev")
    t(f"""
        s_waitcnt lgkmcnt(0)
        v_add_f32 {SUBV}, {diagM}, {SUBV}
        v_min3_f32 {SD}, {SUBV}, {own_i}, {lenv}
        v_mov_b32 {diagM}, {X0}
        v_min3_f32 {Q0}, {SD}, {own_d}, {SHRV}
    """)
    # band-edge cells: MAT.VAL = 100 (b + 1) (src/aln.pyx:502-507) goes in HERE, in front of the record's store (Q0 is one
    # of its data registers); the edge lanes' own compares below then see it -- their traceback word and history record
    # are not stored, their runs are zeroed by edge_fix.  X5 keeps 100 (b + 1) for edge_fix.
    if first or last:
        t(f"""
            v_add_f32 {X5}, 0x42c80000, {O('ev')}
            v_cndmask_b32 {Q0}, {Q0}, {X5}, {O('me') if first and last else O('ml0') if first else O('medge')}
        """)
    # (traceback words: the run registers carry their type tag -- layout.hpp tb_word -- so there is nothing to assemble;
    # 2.0 is the inline constant whose bit pattern is T_LEN << 29)
    if with_len:
        t(f"v_or_b32 {X3}, 2.0, {LENRUN}")
    t(f"""
        v_cmp_eq_f32 vcc, {own_d}, {Q0}
        v_cmp_eq_f32 {O('sa')}, {lenv}, {Q0}
        v_cmp_eq_f32 {O('sb')}, {own_i}, {Q0}
        v_cmp_eq_f32 {O('sc')}, {SUBV}, {Q0}
        v_cndmask_b32 {SE}, {SHRRUN}, {NDELR}, vcc
        v_cndmask_b32 {SE}, {SE}, {X3 if with_len else '2.0'}, {O('sa')}
        v_add_u32 {X3}, {O('hca')}, {O('slot')}
        v_cndmask_b32 {SE}, {SE}, {NINSR}, {O('sb')}
        v_cndmask_b32 {SD}, 0, {DRUN}, {O('sc')}
        v_cndmask_b32 {SE}, {SE}, {DRUN}, {O('sc')}
    """)
    t(f"v_lshl_or_b32 {QRUNS}, {SHRRUN}, 16, {LENRUN}" if with_len else f"v_lshlrev_b32 {QRUNS}, 16, {SHRRUN}")
    t(f"""
        v_lshl_or_b32 {r1}, {NINSR}, 16, {SD}
        v_lshl_or_b32 {r2}, {NDELR}, 16, {SD}
    """)
    # history record (and, where a lane mask is needed anyway, the traceback word) of the band-interior columns; a
    # middle wave holds no others and stores its traceback word behind the hand-over
    if not mid:
        t(f"s_mov_b64 exec, {O('mhist')}")
    t(f"ds_write_b128 {X3}, {QQ}")
    if not mid:
        t(f"global_store_dword {O('tboff')}, {SE}, {O('tbg')}")
        t("s_mov_b64 exec, -1")


def edge_fix(t, first, last):
    """band-edge cells (src/aln.pyx:502-507): the values their one in-band neighbour reads, MAT.VAL aside (mat_part, which
    also left 100 (b + 1) in X5)"""
    own_i, own_d, r1, r2 = O("insv"), O("delv"), O("R1"), O("R2")
    if first and last:
        # a lone wave holds both edges: one mask for the two lanes -- what the other edge's fix writes into an edge cell
        # (DEL of column 2r, INS of column 0) is read by no in-band cell
        t(f"""
            v_cndmask_b32 {own_d}, {own_d}, {X5}, {O('me')}
            v_cndmask_b32 {own_i}, {own_i}, {X5}, {O('me')}
            v_cndmask_b32 {r2}, {r2}, 0, {O('me')}
            v_cndmask_b32 {r1}, {r1}, 0, {O('me')}
        """)
    elif first:
        t(f"""
            v_cndmask_b32 {own_d}, {own_d}, {X5}, {O('ml0')}
            v_cndmask_b32 {r2}, {r2}, 0, {O('ml0')}
        """)
    elif last:
        t(f"""
            v_cndmask_b32 {own_i}, {own_i}, {X5}, {O('medge')}
            v_cndmask_b32 {r1}, {r1}, 0, {O('medge')}
        """)


def next_step(t, mode):
    """loop control of a lone wave: the loop runs to `bend` = the end of the 64-step window, of the span or of what the
    word queues hold, whichever comes first (block_end, out of line): one compare per step"""
    t(f"""
        v_add_f32 {O('ev')}, 0x42c80000, {O('ev')}
        s_add_i32 {O('bl')}, {O('bl')}, 1
        s_cmp_lt_i32 {O('bl')}, {O('bend')}
        s_cbranch_scc0 {L('block_end')}
        s_bitcmp1_b64 {O('mask')}, {O('bl')}
    """)
    if mode == "I":      # (the 'D' body follows)
        t(f"s_cbranch_scc1 {L('mode_i')}")
    else:
        t(f"""
            s_cbranch_scc0 {L('mode_d')}
            s_branch {L('mode_i')}
        """)


def finish(t, mode, first, last, multi, mid, len_variant):
    """behind MAT and the history record: band-edge cells, then (several waves per chunk) the hand-over and the next step.
    The exchange record a neighbour wave will read depends on the NEXT step's kind -- an 'I' step reads the last cell
    of the wave below as its left neighbour, a 'D' step the first cell of the wave above (and the reference words that
    move down with it) as its top neighbour -- and that kind is known here (the loop's own dispatch, done first): only
    that record is written.  At the end of a block of steps (the kind is in another mask word) both are.
    Release: the LDS unit serves the requests of one wave in order, so the record and the history row are in place
    before the progress word that follows them; the explicit lgkmcnt(0) in front of it (like the C++ body's workgroup
    fence) is also what retires the history record's ds_write_b128 before any of its data registers is written again
    (LENST: restored behind it in the variant that follows a LEN candidate)."""
    own_m, own_i, own_d, r1, r2 = MATV, O("insv"), O("delv"), O("R1"), O("R2")
    edge_fix(t, first, last)
    if not multi:
        next_step(t, mode)
        return
    tag = mode + ("L" if len_variant else "")

    def rec63():      # last lane's cell, for the wave above
        t(f"""
            s_mov_b64 exec, {O('ml63')}
            ds_write2_b32 {O('xown')}, {own_m}, {own_d} offset0:{2 * XCH_WORDS} offset1:{2 * XCH_WORDS + 1}
            ds_write2_b32 {O('xown')}, {r2}, {O('seqw')} offset0:{2 * XCH_WORDS + 2} offset1:{2 * XCH_WORDS + 3}
        """)

    def rec0():       # first lane's cell and its reference words, for the wave below (exec = lane 0)
        b = 2 * XCH_WORDS + 5
        t(f"""
            ds_write2_b32 {O('xown')}, {own_m}, {own_i} offset0:{b} offset1:{b + 1}
            ds_write2_b32 {O('xown')}, {r1}, {O('refx')} offset0:{b + 2} offset1:{b + 3}
            ds_write2_b32 {O('xown')}, {O('rc0')}, {O('rc1')} offset0:{b + 4} offset1:{b + 5}
        """)

    def publish():    # (exec = lane 0) the progress word, then the other parity's exchange records
        t(f"""
            s_waitcnt lgkmcnt(0)
            ds_write_b32 {O('progaddr')}, {O('prog')}
            s_mov_b64 exec, -1
        """)
        if len_variant:
            t(f"v_mov_b32 {LENST}, 0x7f800000")
        t(f"""
            v_sub_u32 {O('xown')}, {O('xsum')}, {O('xown')}
            v_sub_u32 {O('xoth')}, {O('xsum')}, {O('xoth')}
        """)
        if mid:
            t(f"global_store_dword {O('tboff')}, {SE}, {O('tbg')}")
        t(f"v_add_f32 {O('ev')}, 0x42c80000, {O('ev')}")

    t(f"""
        v_add_u32 {O('prog')}, 1, {O('prog')}
        s_add_i32 {O('bl')}, {O('bl')}, 1
        s_cmp_lt_i32 {O('bl')}, {O('bend')}
        s_cbranch_scc0 {L('ho_both_' + tag)}
        s_bitcmp1_b64 {O('mask')}, {O('bl')}
        s_cbranch_scc0 {L('ho_d_' + tag)}
    """)
    # the next step is an 'I' step
    if not last:
        rec63()
    t(f"s_mov_b64 exec, {O('ml0')}")
    publish()
    t(f"s_branch {L('mode_i')}")
    # the next step is a 'D' step (behind the 'I' body's main path it falls through into the 'D' body)
    t.label("ho_d_" + tag)
    t(f"s_mov_b64 exec, {O('ml0')}")
    if not first:
        rec0()
    publish()
    if not (mode == "I" and not len_variant):
        t(f"s_branch {L('mode_d')}")
    # the end of a block of steps
    was_rare = t.lines is t.ool
    t.rare()
    t.label("ho_both_" + tag)
    if not last:
        rec63()
    t(f"s_mov_b64 exec, {O('ml0')}")
    if not first:
        rec0()
    publish()
    t(f"s_branch {L('block_end')}")
    if not was_rare:
        t.common()


def tail(t, mode, first, last, multi):
    """MAT, stores, hand-over, next step"""
    mid = multi and not first and not last
    mat_part(t, mode, False, mid, first, last)
    t.label("post_mat_" + mode)
    finish(t, mode, first, last, multi, mid, False)


def block_end(t, first, last):
    """bl has reached bend, the end of a block of steps that needs no test but its own count: the 64-step window, the
    span, and what the word queues (read words entering at column 0: first wave; reference words entering at the last
    column: last wave) and the reference-L window (last wave) hold -- a block is at most as many steps, of either kind, as
    the emptiest of them has entries, so the steps themselves carry no queue tests; near a refill the blocks get short
    (halving), a dozen scalar instructions per block.
    At a window boundary the step window after the new one is fetched, 64 steps before it is needed (the wait also
    drains the traceback stores, once per window) -- also when the span ends there (the caller counts on it).
    blk_setup is also where the text is entered."""
    t.label("block_end")
    t(f"""
        s_and_b32 {O('sx')}, {O('bl')}, 63
        s_cbranch_scc1 {L('rotated')}
        s_mov_b64 {O('mask')}, {O('nmask')}
        v_add_u32 {X3}, {O('kbase')}, {O('laneid')}
        s_add_i32 {O('kbase')}, {O('kbase')}, 64
        global_load_ubyte {X3}, {X3}, {O('stepsg')}
        s_waitcnt vmcnt(0)
        v_cmp_ne_u32 {O('nmask')}, 0, {X3}
    """)
    t.label("rotated")
    t(f"""
        s_cmp_lt_i32 {O('bl')}, {O('b1')}
        s_cbranch_scc0 {L('done')}
    """)
    t.label("blk_setup")
    t(f"""
        s_or_b32 {O('bend')}, {O('bl')}, 63
        s_add_i32 {O('bend')}, {O('bend')}, 1
        s_min_i32 {O('bend')}, {O('bend')}, {O('b1')}
    """)
    if first:
        t(f"""
            s_cmp_ge_i32 {O('sqidx')}, 64
            s_cbranch_scc1 {L('fill_sq')}
        """)
        t.label("sq_ok")
        t(f"""
            s_sub_i32 {O('sx')}, {O('bl')}, {O('sqidx')}
            s_add_i32 {O('sx')}, {O('sx')}, 64
            s_min_i32 {O('bend')}, {O('bend')}, {O('sx')}
        """)
    if last:
        t(f"""
            s_cmp_ge_i32 {O('rqidx')}, 64
            s_cbranch_scc1 {L('fill_rq')}
        """)
        t.label("rq_ok")
        t(f"""
            s_sub_i32 {O('sx')}, {O('bl')}, {O('rqidx')}
            s_add_i32 {O('sx')}, {O('sx')}, 64
            s_min_i32 {O('bend')}, {O('bend')}, {O('sx')}
            s_cmp_ge_i32 {O('sdel')}, {O('dlim')}
            s_cbranch_scc1 {L('fill_win')}
        """)
        t.label("win_ok")
        t(f"""
            s_sub_i32 {O('sx')}, {O('dlim')}, {O('sdel')}
            s_add_i32 {O('sx')}, {O('sx')}, {O('bl')}
            s_min_i32 {O('bend')}, {O('bend')}, {O('sx')}
        """)
    t(f"""
        s_bitcmp1_b64 {O('mask')}, {O('bl')}
        s_cbranch_scc0 {L('mode_d')}
        s_branch {L('mode_i')}
    """)


def xch_reads(t, mode, first, last):
    """the neighbour wave's boundary cell of the previous anti-diagonal (its exchange record of the other parity): an 'I'
    step reads the last cell of the wave below (words 0-3: MAT, DEL, runs, read word), a 'D' step the first cell of the
    wave above (words 5-10: MAT, INS, runs, reference word, the two column descriptors; the descriptors first)"""
    if mode == "I" and not first:
        t(f"ds_read_b128 {XQ}, {O('xoth')}")
    if mode == "D" and not last:
        b = (4 * XCH_WORDS + 5)
        t(f"""
            ds_read2_b32 {X45}, {O('xoth')} offset0:{b + 4} offset1:{b + 5}
            ds_read2_b32 {X01}, {O('xoth')} offset0:{b} offset1:{b + 1}
            ds_read2_b32 {X23}, {O('xoth')} offset0:{b + 2} offset1:{b + 3}
        """)


def poll_issue(t, mode, first, last):
    """the FIRST look at the neighbours' progress words and, right behind it, the read of the neighbour's boundary cell,
    both issued at the head of the step: by the time the work in front of the hand-shake is done they are here.  A
    neighbour that had finished when the look was served has finished now (the words only grow), and the LDS unit
    serves a wave's requests in order, so the cell read behind a look that succeeded is the finished one: the usual
    step then waits for no LDS round trip at the hand-shake.  A look that fails goes on polling out of line and reads
    the cell again."""
    if not ABLATE["nopoll"]:
        if not first and not last:
            t(f"ds_read2_b32 {PP}, {O('pnb')} offset1:2")       # (the neighbours' words lie 8 bytes apart, this wave's in between)
        elif not last:
            t(f"ds_read_b32 {P0}, {O('pnb')} offset:8")
        elif not first:
            t(f"ds_read_b32 {P0}, {O('pnb')}")
    xch_reads(t, mode, first, last)


def polls(t, mode, first, last, sfx):
    """this wave may start the anti-diagonal once its neighbour waves have finished the previous one"""
    if first and last:
        return
    t("s_waitcnt lgkmcnt(0)")
    if ABLATE["nopoll"]:
        return
    both = not first and not last
    again = f"ds_read2_b32 {PP}, {O('pnb')} offset1:2" if both else f"ds_read_b32 {P0}, {O('pnb')}" + (" offset:8" if not last else "")
    test = (f"v_min_i32 {SD}, {P0}, {P1}\nv_cmp_lt_i32 vcc, {SD}, {O('prog')}" if both else f"v_cmp_lt_i32 vcc, {P0}, {O('prog')}")
    t(test)
    t(f"s_cbranch_vccnz {L('pp_first' + sfx)}")
    t.label("pp_ok" + sfx)
    # a wave that has to wait looks again at the LOWEST issue priority: its looks then take no issue slot that a wave of
    # the SIMD with work to do could use (the waves of a chunk share one SIMD, and the one being waited for is among them)
    t.rare()
    t.label("pp_first" + sfx)
    t("s_setprio 0")
    t.label("pp" + sfx)
    t(again)
    t("s_waitcnt lgkmcnt(0)")
    t(test)
    t(f"s_cbranch_vccnz {L('pp' + sfx)}")
    t(f"s_setprio {2 if both else 1}")
    xch_reads(t, mode, first, last)
    t(f"""
        s_waitcnt lgkmcnt(0)
        s_branch {L('pp_ok' + sfx)}
    """)
    t.common()


def book(t, mode):
    """ring row of this anti-diagonal, lane table of history offsets, traceback row"""
    t(f"""
        v_add_u32 {O('slot')}, {O('hw16')}, {O('slot')}
        v_cmp_ne_u32 vcc, {O('ringb')}, {O('slot')}
        v_mov_b32_dpp {O('tab')}, {O('tab')} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
        v_add_u32 {O('tboff')}, {O('tbs4')}, {O('tboff')}
        v_cndmask_b32 {O('slot')}, 0, {O('slot')}, vcc
    """)
    if mode == "I":
        t(f"v_add_u32 {O('tab')}, -16, {O('tab')}")
    t(f"v_cndmask_b32 {O('tab')}, {O('tab')}, {O('slot')}, {O('n0')}")


def gen_role(role):
    multi = role != 0
    first = role in (0, 1)
    last = role in (0, 3)
    mid = role == 2
    t = Text()
    # Nothing the compiled code around this text has in flight may still be on its way when the text starts: the
    # scratch registers are declared clobbered, but the compiler does not wait at an inline-asm statement for loads
    # whose DESTINATION is a clobbered register, and a global load issued by the compiled step that ran just before
    # (the rare-path step the loop hands over) can land in one of them AFTER this text has put its own value there.
    # Seen as a wave whose v100 read as 0 in every lane a few instructions behind the ds_read_b64 that had filled it,
    # once in ~25 full launches with a wave placement that lets a wave issue back to back (LABNOTES round 3); the
    # second read of the same address was right.  One wait per entry (entries = hand-overs, ~1 % of the steps): within
    # the run-to-run spread of the fill time.
    t("s_waitcnt vmcnt(0) lgkmcnt(0)")
    t(f"v_mov_b32 {MATV}, {O('matv')}")
    # The rare-path test of the column descriptors (DSC_RARE), once per entry, over every lane whose descriptor is in
    # the band or on its way there (the lanes beyond the band's last column hold the descriptors that will move into
    # it): an 'I' step does not move the descriptors and a 'D' step tests the one that enters at its last lane before
    # it moves them, so inside the loop no lane ever holds a rare one and a step carries no test of its own
    t(f"""
        v_mov_b32 {LENST}, 0x7f800000
        v_and_b32 {SMR}, 0x80, {O('rc0')}
    """)
    if first and multi:
        t(f"v_cndmask_b32 {SMR}, 0, {SMR}, {O('mhist')}")      # (column 0, an edge: its descriptor moves out)
    t(f"""
        v_cmp_ne_u32 vcc, 0, {SMR}
        s_cbranch_vccnz {L('exit')}
        s_branch {L('blk_setup')}
    """)
    # ================= 'I' step: read words move one column up, "left" is the previous lane.  The column
    # descriptors do not move and INS reads this lane's own cell, so everything that does not depend on the neighbour
    # waves -- the descriptor's summary bits, the lane-table reads, INS -- is done in front of the hand-shake poll.
    t.label("mode_i")
    # (summary bits of the band-interior columns only: the mask 0xbc comes per lane, zero in the lanes of the band's edge
    # columns and beyond -- one instruction where a mask and a select used to be; a middle wave holds no such lanes)
    t(f"v_and_b32 {SMR}, {'0xbc' if mid else O('livebc')}, {O('rc0')}")
    if multi:
        poll_issue(t, "I", first, last)
    book(t, "I")
    shr_tables(t, X3 if first else SD)          # (X3 is waiting for its exchange word)
    t(f"v_add_u32_sdwa {DRUN}, {O('LMr')}, {O('one')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD")
    ins_part(t, "I", X4, X5, O("sb"), (f"v_mov_b32 {O('TMv')}, {MATV}", f"v_mov_b32 {O('TMr')}, {O('R1')}"))
    if first:
        t(f"""
            v_readlane_b32 {O('sx')}, {O('seqq')}, {O('sqidx')}
            s_add_i32 {O('sqidx')}, {O('sqidx')}, 1
        """)
    if multi:
        polls(t, "I", first, last, "_i")
    else:
        t("s_waitcnt lgkmcnt(0)")
    shr_hist(t)
    if not first:
        t(f"""
            v_mov_b32_dpp {X3}, {O('seqw')} wave_shr:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X0}, {MATV} wave_shr:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X1}, {O('delv')} wave_shr:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X2}, {O('R2')} wave_shr:1 row_mask:0xf bank_mask:0xf
            v_mov_b32 {O('seqw')}, {X3}
            v_mov_b32 {O('LMr')}, {X2}
        """)
    else:
        t(f"""
            v_mov_b32_dpp {O('seqw')}, {O('seqw')} wave_shr:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X0}, {MATV} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_mov_b32_dpp {X1}, {O('delv')} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_mov_b32_dpp {X2}, {O('R2')} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_writelane_b32 {O('seqw')}, {O('sx')}, 0
            v_mov_b32 {O('LMr')}, {X2}
        """)
    sub_read(t)
    t("s_waitcnt lgkmcnt(1)")          # the candidate's source record (the substitution score may still be on its way)
    shr_pass(t, mid, "_I", SMR, lambda: del_part(t, "I", X4, X5, O("sb")), lambda: del_part(t, "I", SD, P0, O("sb")), multi)
    len_pass(t, mid, "_I", "I", first, last, multi)
    tail(t, "I", first, last, multi)
    # ================= 'D' step: reference words (and the column descriptors) move one column down, "top" is the
    # next lane.  Last wave of a chunk: the word entering at its last lane comes from its own queue and the next
    # lane's cell is in its own registers, so all but the history reads sits in front of the poll; the other waves
    # learn both from the wave above, behind the poll.
    t.label("mode_d")
    if last:
        t(f"""
            v_readlane_b32 {O('sx')}, {O('rqz')}, {O('rqidx')}
            s_bitcmp1_b32 {O('sx')}, 7
            s_cbranch_scc1 {L('exit')}
        """)
    if multi:
        poll_issue(t, "D", first, last)
    book(t, "D")
    t(f"""
        s_add_i32 {O('sdel')}, {O('sdel')}, 1
        v_add_u32_sdwa {DRUN}, {O('TMr')}, {O('one')} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD
        v_mov_b32 {O('LMv')}, {MATV}
        v_mov_b32 {O('LMr')}, {O('R2')}
    """)
    if last:
        t(f"""
            v_mov_b32_dpp {O('rc0')}, {O('rc0')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X0}, {MATV} wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_writelane_b32 {O('rc0')}, {O('sx')}, 63
            v_readlane_b32 {O('sx')}, {O('rqx')}, {O('rqidx')}
            v_mov_b32_dpp {O('refx')}, {O('refx')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X1}, {O('insv')} wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_mov_b32_dpp {X2}, {O('R1')} wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
            v_writelane_b32 {O('refx')}, {O('sx')}, 63
            v_readlane_b32 {O('sx')}, {O('rqw')}, {O('rqidx')}
            v_mov_b32_dpp {O('rc1')}, {O('rc1')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32 {O('TMr')}, {X2}
            s_add_i32 {O('rqidx')}, {O('rqidx')}, 1
            v_and_b32 {SMR}, {O('livebc')}, {O('rc0')}
            v_writelane_b32 {O('rc1')}, {O('sx')}, 63
        """)
        shr_tables(t)
        ins_part(t, "D", X4, X5, O("sb"))
        if multi:
            polls(t, "D", first, last, "_d")
        else:
            t("s_waitcnt lgkmcnt(0)")
        shr_hist(t)
        sub_read(t)
        t("s_waitcnt lgkmcnt(1)")
    else:
        polls(t, "D", first, last, "_d")
        t(f"""
            v_and_b32 {SD}, 0x80, {X4}
            v_cmp_ne_u32 vcc, 0, {SD}
            s_cbranch_vccnz {L('exit2')}
            v_mov_b32_dpp {X4}, {O('rc0')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32 {O('rc0')}, {X4}
        """)
        shr_tables(t, SD)          # (X3 holds its exchange word)
        t(f"""
            v_and_b32 {SMR}, {'0xbc' if mid else O('livebc')}, {O('rc0')}
            v_mov_b32_dpp {X3}, {O('refx')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X5}, {O('rc1')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X0}, {MATV} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X1}, {O('insv')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32_dpp {X2}, {O('R1')} wave_shl:1 row_mask:0xf bank_mask:0xf
            v_mov_b32 {O('refx')}, {X3}
            v_mov_b32 {O('rc1')}, {X5}
            v_mov_b32 {O('TMr')}, {X2}
        """)
        sub_read(t)
        ins_part(t, "D", X4, X5, O("sb"))      # (in the shadow of the lane-table reads)
        t("s_waitcnt lgkmcnt(1)")
        shr_hist(t)
        t("s_waitcnt lgkmcnt(0)")
    shr_pass(t, mid, "_D", SMR, lambda: del_part(t, "D", X4, X5, O("sb")), lambda: del_part(t, "D", SD, P0, O("sb")), multi)
    len_pass(t, mid, "_D", "D", first, last, multi)
    tail(t, "D", first, last, multi)
    t.lines = t.main
    t.main.extend(t.ool)
    # ---- refills (rare: once per 64 steps of a kind).  They wait for vmcnt(0), which also drains the traceback stores
    if first:
        # read words entering at column 0: the next 64 (SEQW_SENTINEL behind the chunk's last row)
        t.label("fill_sq")
        t(f"""
            s_sub_i32 {O('sqidx')}, {O('sqidx')}, 64
            s_add_i32 {O('sqbase')}, {O('sqbase')}, 64
            v_add_u32 {X3}, {O('sqbase')}, {O('laneid')}
            v_mov_b32 {O('seqq')}, 0xffffc000
            v_cmp_ge_i32 vcc, {O('drows')}, {X3}
            v_lshlrev_b32 {X3}, 2, {X3}
            s_and_saveexec_b64 {O('sa')}, vcc
            global_load_dword {O('seqq')}, {X3}, {O('seqwg')}
            s_waitcnt vmcnt(0)
            s_mov_b64 exec, -1
            s_branch {L('sq_ok')}
        """)
    if last:
        # reference words entering at the last column: the next 64 (REFW_SENTINEL outside the chunk's columns)
        t.label("fill_rq")
        t(f"""
            s_sub_i32 {O('rqidx')}, {O('rqidx')}, 64
            s_add_i32 {O('rqbase')}, {O('rqbase')}, 64
            v_add_u32 {X5}, {O('rqbase')}, {O('laneid')}
            v_mov_b32 {O('rqx')}, 0xdb6d8000
            v_mov_b32 {O('rqz')}, 0
            v_mov_b32 {O('rqw')}, 0
            v_cmp_le_i32 vcc, 0, {X5}
            v_cmp_ge_i32 {O('sa')}, {O('dcols')}, {X5}
            v_lshlrev_b32 {X5}, 4, {X5}
            s_and_b64 vcc, vcc, {O('sa')}
            s_and_saveexec_b64 {O('sa')}, vcc
            global_load_dwordx4 v[92:95], {X5}, {O('refwg')}
            s_waitcnt vmcnt(0)
            v_mov_b32 {O('rqx')}, v92
            v_mov_b32 {O('rqz')}, v94
            v_mov_b32 {O('rqw')}, v95
            s_mov_b64 exec, -1
            s_branch {L('rq_ok')}
        """)
        # reference-L window (LDS): the next WIN_STEP positions, ahead of the band
        ws = 32 if role == 0 else 64
        t.label("fill_win")
        t(f"""
            v_add_u32 {X5}, {O('wfill')}, {O('laneid')}
            v_mov_b32 {P0}, 0
            v_mov_b32 {P1}, 0
            v_cmp_gt_u32 vcc, {ws}, {O('laneid')}
            v_cmp_ge_i32 {O('sa')}, {O('dcols')}, {X5}
            v_lshlrev_b32 {X4}, 3, {X5}
            s_and_b64 {O('sa')}, {O('sa')}, vcc
            s_mov_b64 exec, {O('sa')}
            global_load_dwordx2 {PP}, {X4}, {O('reflg')}
            s_waitcnt vmcnt(0)
            s_mov_b64 exec, vcc
            v_and_b32 {X5}, {O('wmask')}, {X5}
            v_lshlrev_b32 {X5}, 3, {X5}
            v_add_u32 {X5}, {O('winaddr')}, {X5}
            ds_write_b64 {X5}, {PP}
            s_mov_b64 exec, -1
            s_add_i32 {O('wfill')}, {O('wfill')}, {ws}
            s_add_i32 {O('dlim')}, {O('dlim')}, {ws}
            s_branch {L('win_ok')}
        """)
    block_end(t, first, last)
    t.label("exit2")
    t(f"s_mov_b32 {O('status')}, 2")
    t(f"s_branch {L('end')}")
    t.label("exit")
    t(f"s_mov_b32 {O('status')}, 1")
    t(f"s_branch {L('end')}")
    t.label("done")
    t(f"s_mov_b32 {O('status')}, 0")
    t.label("end")
    t(f"v_mov_b32 {O('matv')}, {MATV}")
    # nothing of this text may be in flight when the compiled code resumes: it reuses the scratch registers at once --
    # loads still on their way into them (exit2: the exchange words), and the history record's ds_write_b128, which
    # reads its four data registers for a few cycles after issue
    t("s_waitcnt lgkmcnt(0)")
    return fix_hazards(t.lines)


# ---- hazards -------------------------------------------------------------------------------------------------------
_SREG = re.compile(r"%\[(\w+)\]|\b(vcc|exec)\b")
_VREG = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]|%\[(\w+)\]")


def _regs(tok):
    """registers named by one operand token: set of strings ('v97', '%matv', 'vcc', ...)"""
    tok = tok.strip()
    out = set()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return {"v%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1)}
    if re.fullmatch(r"v\d+", tok):
        return {tok}
    m = re.fullmatch(r"%\[(\w+)\]", tok)
    if m:
        return {"%" + m.group(1)}
    if tok in ("vcc", "exec"):
        return {tok}
    return out


def _split(ins):
    parts = ins.split(None, 1)
    op = parts[0]
    ops = []
    if len(parts) > 1:
        body = re.split(r"\s+(?:offset|offset0|offset1|dst_sel|row_mask|wave_shr|wave_shl|bound_ctrl)\b", parts[1])[0]
        ops = [o.strip() for o in body.split(",")]
    return op, ops


def fix_hazards(lines):
    """Insert s_nop so that (a) a VALU that reads an SGPR / VCC written by a VALU comes at least 2 issue slots later,
    (b) a DPP move reads no VGPR written by a VALU less than 2 slots before, (c) v_readlane reads no VGPR written by
    the previous VALU, (d) v_readlane / v_writelane use no lane-select SGPR written by a VALU within 4 slots.
    Conservative across labels (distances are kept along the fall-through path; a taken branch costs more than these)."""
    out = []
    wrote_s = {}     # SGPR-like name -> index (in `out`) of the VALU that wrote it
    wrote_v = {}     # VGPR-like name -> index of the VALU that wrote it
    for ins in lines:
        if ins.endswith(":"):
            out.append(ins)
            continue
        op, ops = _split(ins)
        is_valu = op.startswith("v_")
        need = 0
        here = len(out)
        if is_valu:
            srcs = ops[1:] if ops else []
            dpp = "wave_sh" in ins or "row_sh" in ins
            for tok in srcs:
                for r in _regs(tok):
                    if r in wrote_s:        # VALU-written SGPR / VCC read by a VALU
                        need = max(need, 2 - (here - wrote_s[r] - 1))
                    if dpp and r in wrote_v:
                        need = max(need, 2 - (here - wrote_v[r] - 1))
                    if op.startswith("v_readlane") and r in wrote_v:
                        need = max(need, 1 - (here - wrote_v[r] - 1))
            if op.startswith("v_readlane") or op.startswith("v_writelane"):
                for r in _regs(ops[-1]):
                    if r in wrote_s:
                        need = max(need, 4 - (here - wrote_s[r] - 1))
            if dpp:                          # (an in-place DPP move also reads its destination)
                for r in _regs(ops[0]):
                    if r in wrote_v:
                        need = max(need, 2 - (here - wrote_v[r] - 1))
        if need > 0:
            out.append("s_nop %d" % (need - 1))
        if is_valu and ops:
            dst = ops[0]
            writes_s = op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane")
            for r in _regs(dst):
                if writes_s:
                    wrote_s[r] = len(out)
                else:
                    wrote_v[r] = len(out)
            if op.startswith("v_cmp") and not dst.startswith("%") and dst != "vcc":
                pass
        elif not is_valu and ops and op.startswith("s_") and not op.startswith("s_cbranch") and not op.startswith("s_waitcnt"):
            for r in _regs(ops[0]):          # rewritten by the scalar unit: no VALU-write hazard left on it
                wrote_s.pop(r, None)
        if op.startswith("ds_read") or op.startswith("global_load") or op.startswith("ds_bpermute"):
            for r in _regs(ops[0]):
                wrote_v.pop(r, None)
        out.append(ins)
    return out


# ---- operand lists ---------------------------------------------------------------------------------------------------
def operands(role):
    multi, first, last = role != 0, role in (0, 1), role in (0, 3)
    outs = [("matv", "+v", "matv"), ("insv", "+v", "insv"), ("delv", "+v", "delv"), ("LMv", "+v", "LMv"), ("TMv", "+v", "TMv"),
            ("R1", "+v", "R1"), ("R2", "+v", "R2"), ("LMr", "+v", "LMr"), ("TMr", "+v", "TMr"), ("seqw", "+v", "seqw"),
            ("refx", "+v", "refx"), ("rc0", "+v", "rc0"), ("rc1", "+v", "rc1"), ("tab", "+v", "env.tab_e"),
            ("slot", "+v", "slot_v"), ("tboff", "+v", "tboff_v"), ("ev", "+v", "e_v"),
            ("bl", "+s", "a_bl"), ("sdel", "+s", "a_sdel"), ("status", "=&s", "a_status"),
            ("mask", "+s", "a_mask"), ("nmask", "+s", "a_nmask"), ("kbase", "+s", "a_kbase"),
            ("sa", "=&s", "a_sa"), ("sb", "=&s", "a_sb"), ("sc", "=&s", "a_sc"), ("bend", "=&s", "a_bend")]
    if multi:
        outs += [("prog", "+v", "prog_v"), ("xown", "+v", "xown"), ("xoth", "+v", "xoth")]
    if first:
        outs += [("sqidx", "+s", "a_sq"), ("sqbase", "+s", "a_sqb"), ("seqq", "+v", "seq_q")]
    if last:
        outs += [("rqidx", "+s", "a_rq"), ("rqbase", "+s", "a_rqb"), ("rqx", "+v", "ref_q.x"), ("rqz", "+v", "ref_q.z"),
                 ("rqw", "+v", "ref_q.w"), ("wfill", "+s", "a_wfill"), ("dlim", "+s", "a_dlim")]
    outs += [("sx", "=&s", "a_sx")]
    ins = [("stepsg", "s", "steps_g"), ("laneid", "v", "a_laneid"), ("b1", "s", "a_b1"), ("hw16", "s", "a_hw16"), ("ringb", "s", "ring_bytes"),
           ("tbs4", "s", "tbstride4"), ("n0", "s", "env.n0_lanes"), ("tbg", "s", "tb_g"), ("istart", "s", "a_istart"),
           ("iext", "s", "a_iext"), ("winaddr", "s", "a_winaddr"), ("wmask", "s", "a_wmask"), ("clampv", "s", "a_clampv"),
           ("clamp1", "s", "a_clamp1"), ("npdim", "s", "a_npdim"), ("gnp", "s", "env.g_np"),
           ("hca", "v", "hist_c_addr"), ("trecip", "v", "env.t_recip"), ("one", "v", "a_one"), ("lanej", "v", "a_lanej"),
           ("inf", "v", "a_inf"), ("c100", "v", "a_c100"),
           ("oneI", "v", "a_oneI"), ("oneD", "v", "a_oneD"), ("tagS", "v", "a_tagS"), ("livebc", "v", "a_livebc")]
    if role != 2:
        ins += [("mhist", "s", "a_mhist")]
    if multi:
        ins += [("xsum", "s", "xsum"), ("pnb", "v", "pnb_addr"), ("progaddr", "v", "a_progaddr"), ("ml0", "s", "a_ml0")]
        if not last:
            ins += [("ml63", "s", "a_ml63")]
    elif first:
        ins += [("ml0", "s", "a_ml0"), ("me", "s", "a_me")]
    if last:
        ins += [("medge", "s", "a_medge"), ("dcols", "s", "a_dcols"), ("refwg", "s", "refw_g"), ("reflg", "s", "refl_g")]
    if first:
        ins += [("drows", "s", "a_drows"), ("seqwg", "s", "seqw_g")]
    return outs, ins


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out_path = os.path.join(here, "fill_step_asm.inc")
    args = sys.argv[1:]
    while args:                      # timing-only ablations: --nopoll / --nolen, with --out FILE
        a = args.pop(0)
        if a in ("--nopoll", "--nolen"):
            ABLATE[a[2:]] = True
        elif a == "--out":
            out_path = args.pop(0)
    out = ["// fill_step_asm.inc -- GENERATED by gen_fill_asm.py (do not edit): the plain-step loop of fill_kernel as gfx950",
           "// assembly, one text per wave role, with the operand lists that bind it to the variables of kernels.hpp.", ""]
    for role in range(4):
        lines = gen_role(role)
        outs, ins = operands(role)
        used = set(re.findall(r"%\[(\w+)\]", "\n".join(lines)))
        declared = {n for n, _, _ in outs + ins}
        assert used <= declared, (role, sorted(used - declared))
        outs = [o for o in outs if o[0] in used or o[1].startswith("+")]
        ins = [i for i in ins if i[0] in used]
        out.append(f"#define NPORE_FILL_ASM_TEXT_{role} \\")
        for ln in lines:
            out.append('    "' + ln + '\\n\\t" \\')
        out.append('    ""')
        out.append(f"#define NPORE_FILL_ASM_OUTS_{role} " + ", ".join(f'[{n}] "{c}"({e})' for n, c, e in outs))
        out.append(f"#define NPORE_FILL_ASM_INS_{role} " + ", ".join(f'[{n}] "{c}"({e})' for n, c, e in ins))
        out.append("")
    out.append("#define NPORE_FILL_ASM_CLOBBERS \"memory\", \"vcc\", \"scc\", " + ", ".join('"%s"' % r for r in SCRATCH))
    with open(out_path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    for role in range(4):
        n = gen_role(role)
        print("role", role, "lines", len(n), "nops", sum(1 for x in n if x.startswith("s_nop")))


if __name__ == "__main__":
    main()
