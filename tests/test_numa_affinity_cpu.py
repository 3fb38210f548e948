"""rank -> GPU -> NUMA-node CPU affinity plan of parallel.py (round 4): pure host logic, no GPU."""
import ldx_amd.parallel as par


def test_parse_cpulist():
    assert par.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert par.parse_cpulist("5") == [5]
    assert par.parse_cpulist("") == []


def test_two_sockets_eight_ranks():
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    allowed = list(range(256))
    node_of_rank = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [par.plan_rank_cpus(allowed, node_of_rank, nodes, r) for r in range(8)]
    for r, cpus in enumerate(got):
        assert len(cpus) == 32 and set(cpus) <= set(nodes[node_of_rank[r]])
    assert len(set().union(*map(set, got))) == 256          # disjoint, everything used


def test_unknown_nodes_and_restricted_affinity():
    # container with 8 allowed CPUs, sysfs says nothing: an even split, never empty
    got = [par.plan_rank_cpus(range(8), [-1] * 3, {}, r) for r in range(3)]
    assert [len(g) for g in got] == [2, 3, 3] and sorted(sum(got, [])) == list(range(8))
    # more ranks than CPUs: everyone still gets one
    assert all(len(par.plan_rank_cpus([4, 5], [-1] * 8, {}, r)) == 1 for r in range(8))
    # node known but none of its CPUs allowed: fall back to the allowed set
    assert par.plan_rank_cpus([0, 1], [1, 1], {1: [64, 65]}, 1) == [1]


def test_masked_ranks_get_disjoint_slices_without_guessing_their_peers(ldx):
    """Per-rank HIP_VISIBLE_DEVICES: a rank knows only its own GPU's node.  Eight ranks spread evenly over two nodes (4 + 4, rank order) each compute their slice
    from (own node, local_rank, local_world) alone: the union covers every core once, no two ranks overlap (the guessed peer map could overlap)."""
    par = ldx.parallel
    nodes = {0: list(range(0, 64)), 1: list(range(64, 128))}
    allowed = list(range(128))
    got = [par.masked_rank_cpus(allowed, 0 if r < 4 else 1, nodes, r, 8) for r in range(8)]
    flat = [c for g in got for c in g]
    assert sorted(flat) == allowed and len(set(flat)) == len(flat) and all(len(g) == 16 for g in got)
    # unknown node: an even cut of what the process may use; never empty, never overlapping
    got = [par.masked_rank_cpus(range(12), -1, {}, r, 4) for r in range(4)]
    assert [len(g) for g in got] == [3, 3, 3, 3] and len({c for g in got for c in g}) == 12
    assert all(len(par.masked_rank_cpus([7], -1, {}, r, 8)) == 1 for r in range(8))


def test_run_nonce_without_the_launcher_variable_names_the_torchrun_launch(ldx, monkeypatch):
    par = ldx.parallel
    monkeypatch.delenv("LDX_RCCL_NONCE", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", "29511"); monkeypatch.setenv("TORCHELASTIC_RUN_ID", "a")
    a = par._run_nonce()
    monkeypatch.setenv("MASTER_PORT", "29512")
    b = par._run_nonce()
    monkeypatch.setenv("LDX_RCCL_NONCE", "x")
    c = par._run_nonce()
    assert len({a, b, c}) == 3 and a == par._run_nonce("127.0.0.1|29511|a|") and par._run_nonce("x") == c
