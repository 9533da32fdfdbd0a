"""Where the weight-gradient kernel's LDS bank conflicts come from -- computed from the address formulas of csrc/wgrad_body.h with the
banking rules of /opt/skills/guides/MI355X_MICROARCH.md section "LDS [CDNA4]" (DESIGN.md 2.5).

PMC of the fine launch (profiles/r04_pmc_traffic_v1.json): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 7.7 %.  DESIGN.md used to call the
kernel "bank-conflict-free"; that holds for its transposed reads (every plain job: 6 of 9), not for the two recomputing jobs:

  layer_7 job   16 x ds_read_b32 of the mask words per tile: lanes of feature half hf = 0 / 1 read addresses 512 B apart = the same bank
                (banks repeat every 128 B for ds_read_b32): 2-way;  1 x ds_read_b128 of the dz_head block row-wise: 2-way
  layer_1 job   4 x ds_read_b128 of the enc blocks row-wise (lane = sample, 32 B per sample: layout.h saved_off): 2-way

The model below reproduces the split (and the total to within a point of the counter); the per-variant counters of the ablation
builds (-DKNERF_WGRAD_ABLATE_LDS=1|2|3) are in profiles/r05_wgrad_lds_conflicts.json.  A mask block with a sample's two halves adjacent
(-DKNERF_MASK_LAYOUT=1) makes the mask reads conflict-free (7.7 % -> 3.1 % by PMC, profiles/r05_wgrad_lds_conflicts_mask_layout.json) and
was 0.4-1.0 % slower per fine launch in every pairing (profiles/r05_mask_layout_ab.json): the product keeps lane * 16."""
import numpy as np

# lane groups (one LDS cycle each when conflict-free) and bank modulus per instruction: MI355X_MICROARCH.md, LDS table
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(32)), list(range(32, 64))]
RULES = {"ds_read_b32": (HALVES, 32, 4), "ds_read_b64_tr_b16": (HALVES, 64, 8), "ds_read_b128": (B128_GROUPS, 64, 16)}


def cycles(instr, addr):
    """(LDS-array cycles, of which conflict cycles) of one wave-instruction: per lane group, the largest number of DISTINCT
    addresses that fall on one bank (identical addresses broadcast)"""
    groups, mod, width = RULES[instr]
    total = extra = 0
    for g in groups:
        on_bank = {}
        for lane in g:
            for b in range(width // 4):
                on_bank.setdefault(((addr[lane] + 4 * b) // 4) % mod, set()).add(addr[lane])
        worst = max(len(v) for v in on_bank.values())
        total += worst; extra += worst - 1
    return total, extra


LANES = np.arange(64)
GRP, IL = LANES >> 4, LANES & 15
PAR, H, Q, P = GRP & 1, GRP >> 1, IL >> 2, IL & 3


def tr_plain(r):          # wgrad_job_body lane_off[r]
    return PAR * 1024 + (2 * (8 * H + 4 * (r ^ PAR) + Q) + (P & 1)) * 16 + (P >> 1) * 8


def tr_permuted(r):       # wgrad_l1_recompute / wgrad_last_recompute lane_off[r]
    return PAR * 1024 + (2 * (4 * (H ^ PAR) + 8 * r + Q) + (P & 1)) * 16 + (P >> 1) * 8


ROW = (2 * (LANES & 31) + (LANES >> 5)) * 16          # enc_off / zoff: saved_off(b even, h = lane >> 5, s = lane & 31)
LINEAR = LANES * 16


def mask_addr(r, wo, layout=0):     # dz7_mfma: word of (sample row of accumulator register r, feature half of this lane's column)
    """layout 0 (the product, csrc/layout.h mask_lane_off): lane * 16, i.e. a sample's two feature halves 512 B apart;
    layout 1 (-DKNERF_MASK_LAYOUT=1, the round-5 experiment): the halves adjacent, 32 B per sample"""
    c, hh = LANES & 31, LANES >> 5
    hf = (c >> 2) & 1
    smp = (r & 3) + 8 * (r >> 2) + 4 * hh
    return 1024 + ((2 * smp + hf) * 16 if layout else (hf * 32 + smp) * 16) + (wo >> 1) * 4


def test_transposed_reads_are_conflict_free_in_every_job():
    for f in (tr_plain, tr_permuted):
        for r in range(2):
            for kk in range(2):
                for odd_block in (0, 1):
                    assert cycles("ds_read_b64_tr_b16", f(r) + kk * 512 + odd_block * 0) == (2, 0)
    assert cycles("ds_read_b128", LINEAR) == (4, 0)                        # the h0 fragments of the layer_1 job (xr + lane * 16)


def test_row_wise_b128_reads_of_sample_major_blocks_are_two_way():
    for ks in range(4):
        assert cycles("ds_read_b128", ROW ^ ((ks & 1) << 7)) == (8, 4)     # enc blocks (layer_1), dz_head block (layer_7)


def test_mask_word_reads_are_two_way_and_would_be_conflict_free_with_an_interleaved_mask_block():
    for r in range(16):
        for wo in range(8):
            assert cycles("ds_read_b32", mask_addr(r, wo, layout=0)) == (4, 2)          # 512 B apart = the same bank of 32
            assert cycles("ds_read_b32", mask_addr(r, wo, layout=1)) == (2, 0)          # KNERF_MASK_LAYOUT=1: 16 B apart


def _modelled_share(layout):
    """LDS-array cycles per 32-sample tile, summed over the eight waves of a workgroup, job by job (each tile goes through all nine
    jobs of the default shape): the conflict share is what SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measures."""
    tr = cycles("ds_read_b64_tr_b16", tr_plain(0))[0]                       # 2
    frag = 2 * tr                                                          # tr_frag = two transposed reads

    def plain(n_acc_per_wave):                                             # per wave and tile: 2 k-steps x (1 dz + NACC input fragments)
        return 8 * 2 * (1 + n_acc_per_wave) * frag
    total = {"layer_0": plain(3), "head": plain(2), "layer_5": plain(11)}
    total.update({f"layer_{l}": plain(9) for l in (2, 3, 4, 6)})
    conflicts = dict.fromkeys(total, 0)
    row_t, row_x = cycles("ds_read_b128", ROW)
    lin_t, _ = cycles("ds_read_b128", LINEAR)
    m_t, m_x = cycles("ds_read_b32", mask_addr(0, 0, layout))
    # layer_1: 2 k-steps x 1 dz fragment, 16 h0 fragments (b128, linear), 4 enc rows (b128), 2 ds_write_b128 (8 LDS-array cycles each)
    total["layer_1"] = 8 * (2 * frag + 16 * lin_t + 4 * row_t + 2 * 8); conflicts["layer_1"] = 8 * 4 * row_x
    # layer_7: 2 k-steps x 8 h6 fragments, 1 dz_head row (b128), 16 mask words (b32)
    total["layer_7"] = 8 * (2 * 8 * frag + row_t + 16 * m_t); conflicts["layer_7"] = 8 * (row_x + 16 * m_x)
    return sum(conflicts.values()) / sum(total.values()), 8 * 16 * m_x / sum(conflicts.values())


def test_modelled_conflict_share_of_a_launch_matches_the_counter():
    share, mask_part = _modelled_share(layout=0)
    assert 0.06 < share < 0.09, share                                      # the counter: 0.0766
    assert 0.55 < mask_part < 0.70                                         # PMC on the ablation builds: 61.5 % of the conflict cycles are the mask words
    share, mask_part = _modelled_share(layout=1)
    assert 0.02 < share < 0.04 and mask_part == 0.0, share                 # PMC of the KNERF_MASK_LAYOUT=1 build: 0.0309 (the row-wise b128 reads of enc / dz_head)


def test_general_shape_cooperative_wgrad_swizzle_is_conflict_free():
    """csrc/generic.hip wgrad_coop_kernel (round 5): unpadded 512-byte rows staged by LDS-DMA, the 16-byte chunks of row r at
    chunk ^ 4 (r & 3); the transposed reads of every (tile, k-step, first / second read) touch four rows whose windows then cover the
    64 banks exactly once -- and WITHOUT the swizzle the same reads would be 4-way (every row starts at bank 0)"""
    gq, il = LANES >> 4, LANES & 15
    tq = il >> 2
    base = (8 * (gq >> 1) + tq) * 512 + (2 * (gq & 1) + ((il & 3) >> 1)) * 16 + (il & 1) * 8
    for tile in range(8):
        for kk in range(2):
            for second in (0, 1):
                swz = base + (16 * kk + 4 * second) * 512 + ((tile ^ tq) << 6)
                assert cycles("ds_read_b64_tr_b16", swz) == (2, 0)
                plain = base + (16 * kk + 4 * second) * 512 + (tile << 6)
                assert cycles("ds_read_b64_tr_b16", plain) == (8, 6)
    # the staging side: slot (lane & 31) of row 2m + (lane >> 5) receives chunk slot ^ 4 (row & 3): a permutation of the row's 32 chunks
    for m in range(16):
        for hh in range(2):
            row = 2 * m + hh
            assert sorted(int(x) ^ ((row & 3) << 2) for x in range(32)) == list(range(32))
