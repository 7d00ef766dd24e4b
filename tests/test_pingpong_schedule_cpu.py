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
    # the peeled LEAN tail: steady state while tile t+3 exists, then nk-3 (group 0 only requests), nk-2 and nk-1 (nobody requests, vmcnt(0))
    assert 'for (; t + 3 < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});' in body
    assert 'if (nk >= 3) { ktile(t, T_{}, T_{}, F_{}, F_{}); ++t; }' in body
    assert 'ktile(t, F_{}, F_{}, F_{}, F_{}); ++t;' in body and 'ktile(t, F_{}, F_{}, F_{}, T_{});' in body
    assert 'for (int t = 0; t < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});' in body                  # non-LEAN: every k-tile alike
    assert 'if (grp) __builtin_amdgcn_s_barrier();' in body and 'if (!LEAN && !grp) __builtin_amdgcn_s_barrier();' in body
    assert 'if constexpr (LAST) { if (!grp) __builtin_amdgcn_s_barrier(); }' in body
    assert body.count('__builtin_amdgcn_s_barrier()') == 6                        # B0, skew, end of R, end of M (two forms), trailing


# ---------------------------------------------------------------------------------------------------------------------------
# Wave-specialised form (big_tile_ws, round 4): four producer waves issue every LDS-DMA and own the vmcnt bookkeeping, the eight
# consumer waves only read LDS, multiply and meet the barriers.  Three actors: consumer group 0, consumer group 1, the producers.
WS_DMA_PER_TILE = 12      # instructions per PRODUCER wave and k-tile (8 A + 4 B)


def schedule_ws(nk, actor):
    if actor == 'p':
        ev = [('issue', 0, 0)]
        if nk > 1:
            ev += [('issue', 1, 1)]
        if nk > 2:
            ev += [('issue', 2, 2)]
        ev += [('wait_vm', (min(nk, 3) - 1) * WS_DMA_PER_TILE), ('barrier',)]             # B0: tile 0 landed
        st2 = 2
        for t in range(nk):
            req = t >= 1 and t + 2 < nk
            if req:
                ev += [('issue_half', t + 2, st2, 0)]                                     # first six requests of tile t+2 in phase 2t ...
            ev += [('barrier',)]                                                          # end of phase 2t
            if req:
                ev += [('issue_half', t + 2, st2, 1)]                                     # ... the other six in phase 2t+1, in front of the wait
            ev += [('wait_vm', WS_DMA_PER_TILE if t + 2 < nk else 0), ('barrier',)]        # end of phase 2t+1: tile t+1 landed
            st2 = (st2 + 1) % 3
        ev += [('barrier',)]                                                              # the barrier behind the parked C tile
        return ev
    ev = [('barrier',)]
    if actor:
        ev += [('barrier',)]                      # group 1 idles through phase 0
    st = 0
    for t in range(nk):
        ev += [('read', t, st), ('wait_lgkm',), ('barrier',)]
        if not (actor and t == nk - 1):
            ev += [('barrier',)]
        st = (st + 1) % 3
    ev += [('park',), ('barrier',)]
    return ev


def run_ws(nk, mutate=None):
    actors = [0, 1, 'p']
    evs = {a: schedule_ws(nk, a) for a in actors}
    if mutate:
        evs = mutate(evs)
    nb = {a: sum(e[0] == 'barrier' for e in evs[a]) for a in actors}
    assert len(set(nb.values())) == 1, nb                                                 # barrier parity of all twelve waves
    pos = {a: 0 for a in actors}
    outstanding, pending_pub, published, landed_halves = [], set(), set(), set()
    reads_inflight = {0: [], 1: []}
    retired_unpub = {0: set(), 1: set()}
    busy = {0: set(), 1: set(), 2: set()}
    content = {0: None, 1: None, 2: None}
    parked = {0: False, 1: False}
    while any(pos[a] < len(evs[a]) for a in actors):
        for a in actors:
            while pos[a] < len(evs[a]):
                e = evs[a][pos[a]]; pos[a] += 1
                if e[0] == 'issue':
                    _, tile, stage = e
                    assert not any(parked.values()), f'nk={nk}: tile {tile} requested after a C tile was parked in the ring'
                    assert not busy[stage], f'nk={nk}: tile {tile} overwrites stage {stage} while groups {busy[stage]} may read it'
                    assert stage == tile % 3, (nk, tile, stage)
                    outstanding += [(tile, stage, 0), (tile, stage, 1)]; content[stage] = tile
                elif e[0] == 'issue_half':
                    _, tile, stage, h = e
                    assert not any(parked.values()), f'nk={nk}: tile {tile} requested after a C tile was parked in the ring'
                    assert not busy[stage], f'nk={nk}: tile {tile} overwrites stage {stage} while groups {busy[stage]} may read it'
                    assert stage == tile % 3, (nk, tile, stage)
                    outstanding.append((tile, stage, h)); content[stage] = tile
                elif e[0] == 'wait_vm':
                    keep = e[1] // (WS_DMA_PER_TILE // 2)                                  # the counter counts instructions: six per half tile
                    done, outstanding = outstanding[:len(outstanding) - keep], outstanding[len(outstanding) - keep:]
                    landed_halves |= set(done)
                    pending_pub |= {(t_, s_) for (t_, s_, h_) in done if (t_, s_, 1 - h_) in landed_halves}
                elif e[0] == 'read':
                    _, tile, stage = e
                    assert (tile, stage) in published, f'nk={nk}: group {a} reads tile {tile} before it is published'
                    assert content[stage] == tile, (nk, a, tile, stage, content[stage])
                    reads_inflight[a].append(stage); busy[stage].add(a)
                elif e[0] == 'wait_lgkm':
                    retired_unpub[a] |= set(reads_inflight[a]); reads_inflight[a] = []
                elif e[0] == 'park':
                    assert not outstanding, f'nk={nk}: group {a} parks the C tile with DMA in flight {outstanding}'
                    for stage in (0, 1, 2):
                        assert not busy[stage], f'nk={nk}: group {a} parks over stage {stage} while {busy[stage]} may read it'
                    parked[a] = True
                elif e[0] == 'barrier':
                    break
        published |= pending_pub; pending_pub = set()
        for a in (0, 1):
            for stage in retired_unpub[a]:
                busy[stage].discard(a)
            retired_unpub[a] = set()
    assert not outstanding and all(parked.values())
    assert {t for t, _ in published} == set(range(nk))                                    # every k-tile was requested exactly as needed


@pytest.mark.parametrize('nk', list(range(2, 41)) + [48, 96, 620])
def test_wave_specialised_ring_has_no_raw_or_war_hazard(nk):
    run_ws(nk)


def test_ws_model_detects_an_early_request_and_a_shallow_wait():
    def early(evs):                                # request tile t+2 one phase early (phase 2t-1): group 1 may still read tile t-1
        ev = list(evs['p'])
        i = next(k for k, e in enumerate(ev) if e[0] == 'issue_half' and e[1] == 3)
        e = ev.pop(i)
        j = max(k for k in range(i) if ev[k][0] == 'barrier')          # in front of the previous barrier
        ev.insert(j, e)
        return dict(evs, p=ev)
    with pytest.raises(AssertionError):
        run_ws(8, early)

    def shallow(evs):                              # allow two tiles in flight where one is the limit: tile t+1 read before it landed
        return dict(evs, p=[('wait_vm', 2 * WS_DMA_PER_TILE) if (e[0] == 'wait_vm' and e[1] == WS_DMA_PER_TILE) else e for e in evs['p']])
    with pytest.raises(AssertionError):
        run_ws(8, shallow)


def test_ws_model_constants_match_the_kernel_source():
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc', 'gemm_big.hip')).read()
    body = src[src.index('__device__ __forceinline__ void big_tile_ws('):src.index('__global__ __launch_bounds__(NTHR_WS) void gemm_big_ws_kernel')]
    assert 'constexpr int A_PER = BM / 8 / 4, B_PER = BN / 8 / 4;' in body and (256 // 8 + 128 // 8) // 4 == WS_DMA_PER_TILE
    assert 'if (nk > 2) asm volatile("s_waitcnt vmcnt(24)"' in body and 'else if (nk > 1) asm volatile("s_waitcnt vmcnt(12)"' in body
    assert 'const bool req = t >= 1 && t + 2 < nk;' in body and 'int st2 = 2;' in body
    loop = body[body.index('for (int t = 0; t < nk; ++t) {'):]
    h0, b1, h1, w = (loop.index('request_half(t + 2, st2, std::integral_constant<int, 0>{})'), loop.index('__builtin_amdgcn_s_barrier();'),
                     loop.index('request_half(t + 2, st2, std::integral_constant<int, 1>{})'), loop.index('s_waitcnt vmcnt(12)'))
    assert h0 < b1 < h1 < w                                          # half 0 | barrier | half 1, wait | barrier
    assert 'if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");\n      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");' in body
    prod = body[body.index('if (wave >= 8) {'):body.index('// ==================================================================== consumers')]
    assert prod.count('__builtin_amdgcn_s_barrier()') == 4          # B0, two per k-tile, the park barrier
    cons = body[body.index('// ==================================================================== consumers'):]
    assert 'vmcnt' not in cons.split('// ---- epilogue')[0]          # consumers carry no memory waits in the k loop
    assert 'if constexpr (LAST) { if (!grp) __builtin_amdgcn_s_barrier(); }' in cons
