"""Model check of the ping-pong main loop of the 256x128 LDS-DMA GEMM kernel (prismer_amd/csrc/gemm_big.hip, VARIANT & 4).

The kernel's correctness rests on an ordering argument between LDS-DMA writes (global_load_lds, retired by counted s_waitcnt vmcnt),
LDS reads (ds_read, retired by lgkmcnt(0)) and workgroup barriers; a mis-count shows up on hardware only as a rare wrong tile.  This
test re-states the schedule as data -- which group does what in which phase, where the waits and barriers are -- and checks, for
every k-tile count, the two hazards on every stage of the 3-stage ring:
  RAW  a group reads tile t from stage t % 3 only after EVERY wave's DMA of tile t has been waited for before a barrier that precedes
       the reading phase;
  WAR  a DMA into a stage is issued only after every read of the tile previously held by that stage has been retired (lgkmcnt(0))
       before a barrier that precedes the issuing phase.
The GPU tests (tests/test_kernels_gpu.py::test_gemm_big_tile_lds_dma_kernel, repeated launches) check the implementation; this checks
the argument, including the clamped surplus loads of the last iterations and the one-phase skew between the two wave groups."""
import pytest

DMA_PER_TILE = 6          # instructions per thread and k-tile (4 A + 2 B); 3 for a 32-wide stage


def schedule(nk, group, lean=False):
    """events of one wave of `group` in program order: ('issue', tile, stage) | ('wait_vm', n_outstanding_allowed) | ('wait_lgkm',)
    | ('read', tile, stage) | ('barrier',).  Mirrors the source: prologue, then per k-tile R(t) and M(t), then the tail.
    lean (VARIANT & 8, round 4): tiles that do not exist are not requested (no clamped surplus loads, nothing to drain), a wave whose
    newest request is the tile it needs waits vmcnt(0), group 1 skips the barrier behind its last M phase and group 0 does not idle."""
    ev = [('issue', 0, 0), ('issue', min(1, nk - 1), 1)]
    if group:
        if not lean or nk > 2:
            ev += [('issue', min(2, nk - 1), 2), ('wait_vm', 2 * DMA_PER_TILE)]
        else:
            ev += [('wait_vm', DMA_PER_TILE)]
    else:
        ev += [('wait_vm', DMA_PER_TILE)]
    ev += [('barrier',)]
    if group:
        ev += [('barrier',)]                      # group 1 idles through phase 0
    st = 0
    for t in range(nk):
        newer = (not lean) or t + 2 < nk
        ev += [('read', t, st)]
        if group:
            ev += [('wait_vm', DMA_PER_TILE if newer else 0)]
        ev += [('wait_lgkm',), ('barrier',)]
        if not lean or t + 2 + group < nk:
            ev += [('issue', min(t + 2 + group, nk - 1), (st + 2 + group) % 3)]
        if not group:
            ev += [('wait_vm', DMA_PER_TILE if newer else 0)]
        if not (lean and group and t == nk - 1):
            ev += [('barrier',)]
        st = (st + 1) % 3
    if not lean:
        if not group:
            ev += [('barrier',)]                  # group 0 idles through the last phase
        ev += [('wait_vm', 0), ('barrier',)]
    ev += [('park',), ('barrier',)]               # epilogue: the C tile is written over the ring, then one barrier before it is read back
    return ev


def run(nk, lean=False):
    evs = [schedule(nk, 0, lean), schedule(nk, 1, lean)]
    assert sum(e[0] == 'barrier' for e in evs[0]) == sum(e[0] == 'barrier' for e in evs[1])     # barrier parity
    pos = [0, 0]
    outstanding = [[], []]            # per group: DMA issued, not yet known landed: (tile, stage), program order (vmcnt retires in order)
    landed_known = [set(), set()]     # (tile, stage) whose landing this group has waited for
    published = set()                 # (tile, stage, group): landing waited for AND a barrier passed since
    pending_pub = [set(), set()]
    reads_inflight = [[], []]         # per group: stages with reads not yet retired
    reads_retired_unpublished = [set(), set()]
    busy_reads = {0: set(), 1: set(), 2: set()}      # stage -> groups whose reads of the CURRENT content may still be in flight
    content = {0: None, 1: None, 2: None}            # stage -> tile most recently DMA-issued into it (by anyone)
    parked = [False, False]
    while pos[0] < len(evs[0]) or pos[1] < len(evs[1]):
        # run each group up to (and including) its next barrier, then release both: a barrier is a phase boundary for everyone
        for g in (0, 1):
            while pos[g] < len(evs[g]):
                e = evs[g][pos[g]]; pos[g] += 1
                if e[0] == 'issue':
                    _, tile, stage = e
                    assert not any(parked), f'nk={nk}: group {g} requests tile {tile} after a group has parked its C tile in the ring'
                    # WAR: nobody may still be reading what this stage held (reads retired AND published by a barrier)
                    assert not busy_reads[stage], f'nk={nk}: group {g} overwrites stage {stage} (tile {tile}) while groups {busy_reads[stage]} may read it'
                    outstanding[g].append((tile, stage))
                    content[stage] = tile
                elif e[0] == 'wait_vm':
                    keep = e[1] // DMA_PER_TILE
                    done, outstanding[g] = outstanding[g][:len(outstanding[g]) - keep], outstanding[g][len(outstanding[g]) - keep:]
                    pending_pub[g] |= set(done)
                elif e[0] == 'read':
                    _, tile, stage = e
                    # RAW: both groups' shares of this tile landed and were published by a barrier this group has passed
                    assert (tile, stage, 0) in published and (tile, stage, 1) in published, \
                        f'nk={nk}: group {g} reads tile {tile} from stage {stage} before both DMA shares are published'
                    assert content[stage] == tile or content[stage] == min(tile, nk - 1), (nk, g, tile, stage, content[stage])
                    reads_inflight[g].append(stage)
                    busy_reads[stage].add(g)
                elif e[0] == 'park':
                    # the ring becomes the C tile: nothing of THIS wave may still be landing, and nobody may still read any stage
                    assert not outstanding[g], f'nk={nk}: group {g} parks the C tile with DMA in flight {outstanding[g]}'
                    for stage in (0, 1, 2):
                        assert not busy_reads[stage], f'nk={nk}: group {g} parks over stage {stage} while groups {busy_reads[stage]} may read it'
                    parked[g] = True
                elif e[0] == 'issue' and any(parked):
                    raise AssertionError('DMA after a park')
                elif e[0] == 'wait_lgkm':
                    reads_retired_unpublished[g] |= set(reads_inflight[g]); reads_inflight[g] = []
                elif e[0] == 'barrier':
                    break
        # barrier release: what each group waited for before it becomes visible to everyone
        for g in (0, 1):
            for (tile, stage) in pending_pub[g]:
                published.add((tile, stage, g))
            pending_pub[g] = set()
            for stage in reads_retired_unpublished[g]:
                busy_reads[stage].discard(g)
            reads_retired_unpublished[g] = set()
    assert not outstanding[0] and not outstanding[1]            # ring drained before it is reused as the C tile
    assert all(parked)


@pytest.mark.parametrize('lean', [False, True])
@pytest.mark.parametrize('nk', list(range(2, 41)) + [48, 96, 620])
def test_pingpong_ring_has_no_raw_or_war_hazard(nk, lean):
    run(nk, lean)


def test_model_detects_a_park_under_a_reader():
    """sanity of the park check: without the lgkmcnt(0) of the last R phase a group would park over a stage that is still read"""
    import tests.test_pingpong_schedule_cpu as me
    orig = me.schedule

    def bad(nk, group, lean=False):
        ev = orig(nk, group, lean)
        last = max(i for i, e in enumerate(ev) if e[0] == 'wait_lgkm')
        return ev[:last] + ev[last + 1:]
    try:
        me.schedule = bad
        with pytest.raises(AssertionError):
            me.run(8, True)
    finally:
        me.schedule = orig


def test_model_detects_a_too_shallow_wait():
    """sanity of the checker itself: waiting for one tile less (vmcnt(12) where vmcnt(6) is needed) must be flagged"""
    global DMA_PER_TILE
    good = schedule(8, 0)
    bad = [('wait_vm', 2 * DMA_PER_TILE) if (e[0] == 'wait_vm' and e[1] == DMA_PER_TILE) else e for e in good]
    import tests.test_pingpong_schedule_cpu as me
    orig = me.schedule
    try:
        me.schedule = lambda nk, group, lean=False: bad if group == 0 else orig(nk, 1)
        with pytest.raises(AssertionError):
            me.run(8)
    finally:
        me.schedule = orig


def test_model_constants_match_the_kernel_source():
    """the model above and gemm_big.hip must describe the same schedule: DMA instructions per tile, the counted waits, the tile each
    group issues in its M phase, the skew barrier of group 1 and the trailing one of group 0"""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc', 'gemm_big.hip')).read()
    m = re.search(r'constexpr int A_INSTR = BM \* 8 / NTHR, B_INSTR = BN \* 8 / NTHR;', src)
    assert m and 'constexpr int NTHR = 512' in src
    assert (256 * 8 + 128 * 8) // 512 == DMA_PER_TILE
    body = src[src.index('// ---- ping-pong main loop'):src.index('drain the surplus DMA')]
    assert body.count(f'"s_waitcnt vmcnt({DMA_PER_TILE})"') == 3                  # group 0 prologue, group 1 prologue (LEAN, nk == 2), group 0 after M(t)
    assert body.count(f'"s_waitcnt vmcnt({DMA_PER_TILE}) lgkmcnt(0)"') == 1       # group 1 at the end of R(t)
    assert body.count('"s_waitcnt vmcnt(0) lgkmcnt(0)"') == 1 and body.count('"s_waitcnt vmcnt(0)"') == 2     # the same two waits without a newer tile (LEAN) + the non-LEAN drain
    assert body.count(f'"s_waitcnt vmcnt({2 * DMA_PER_TILE})"') == 1              # group 1 prologue (three tiles issued)
    assert 'if constexpr (ISS0 && ISS1) issue(min(t + 2 + grp, nk - 1), sn);' in body and 'int sn = st + 2 + grp' in body
    assert 'else if constexpr (ISS0) { if (!grp) issue(min(t + 2, nk - 1), sn); }' in body
    # SPREAD (round 6): the same requests of the same tile into the same stage, one piece behind every second MFMA of the M phase
    assert 'const bool mine = ISS0 && (ISS1 || !grp);' in body and 'const int kt_n = min(t + 2 + grp, nk - 1);' in body
    assert 'if (mine) issue_piece(kt_n, sn, std::integral_constant<int, m / 2>{});' in body
    assert 'if constexpr (ISS0 && (m & 1) && m / 2 < A_INSTR + B_INSTR)' in body
    # the peeled LEAN tail: steady state while tile t+3 exists, then nk-3 (group 0 only requests), nk-2 and nk-1 (nobody requests, vmcnt(0))
    assert 'for (; t + 3 < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});' in body
    assert 'if (nk >= 3) { ktile(t, T_{}, T_{}, F_{}, F_{}); ++t; }' in body
    assert 'ktile(t, F_{}, F_{}, F_{}, F_{}); ++t;' in body and 'ktile(t, F_{}, F_{}, F_{}, T_{});' in body
    assert 'for (int t = 0; t < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});' in body                  # non-LEAN: every k-tile alike
    assert 'if (grp) __builtin_amdgcn_s_barrier();' in body and 'if (!LEAN && !grp) __builtin_amdgcn_s_barrier();' in body
    assert 'if constexpr (LAST) { if (!grp) __builtin_amdgcn_s_barrier(); }' in body
    assert body.count('__builtin_amdgcn_s_barrier()') == 6                        # B0, skew, end of R, end of M (two forms), trailing
