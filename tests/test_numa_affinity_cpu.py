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
