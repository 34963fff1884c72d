"""HIP FRNN vs the oracle's exact brute force.  Bar: bit-exact neighbour indices and
squared distances (integer / f32 with contraction off on both sides)."""
import pytest
import torch

from util import cube_cloud, sphere_cloud

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import iso_oracle
    return iso_oracle


def _cmp(dev, p1, p2, l1, l2, K, r, return_nn=True, same=False):
    O = _oracle()
    from iso_points_amd import frnn
    d_ref, i_ref, nn_ref, _ = O.frnn_grid_points(p1, p2, l1, l2, K=K, r=r, return_nn=return_nn)
    g2 = p2.to(dev)
    g1 = g2 if same else p1.to(dev)
    gl1 = None if l1 is None else l1.to(dev)
    gl2 = gl1 if (same and l1 is l2) else (None if l2 is None else l2.to(dev))
    rr = r.to(dev) if torch.is_tensor(r) else r
    d, i, nn, grid = frnn.frnn_grid_points(g1, g2, gl1, gl2, K=K, r=rr, return_nn=return_nn)
    assert i.dtype == torch.int64 and d.dtype == torch.float32
    assert torch.equal(i.cpu(), i_ref), "neighbour indices differ"
    assert torch.equal(d.cpu(), d_ref), "squared distances differ"
    if return_nn:
        assert torch.equal(nn.cpu(), nn_ref)
    return d, i, nn, grid


@pytest.mark.parametrize("P,K,knn_k", [(2000, 9, 8), (5000, 17, 16), (300, 7, 6)])
def test_self_query_tree_radius(dev, P, K, knn_k):
    """_create_tree (levelset_sampling.py:129-138): r = sqrt(diag/P)*knn_k, self included."""
    O = _oracle()
    p = sphere_cloud(P, seed=P)
    r = O.search_radius(p, torch.tensor([P]), knn_k)
    d, i, nn, _ = _cmp(dev, p, p, None, None, K, r, same=True)
    assert (i[0, :, 0].cpu() == torch.arange(P)).all()  # self is the nearest (d2 = 0)


def test_large_radius_dense(dev):
    """rasterizer.py:371 shape: K=7, r=0.2 on a dense cloud (ring search stops early)."""
    p = sphere_cloud(20000, seed=11)
    _cmp(dev, p, p, None, None, 7, 0.2, return_nn=False, same=True)


def test_small_radius_few_hits_and_padding(dev):
    p = cube_cloud(3000, seed=2)
    d, i, _, _ = _cmp(dev, p, p, None, None, 9, 0.03, same=True)
    assert (i == -1).any() and (d[i == -1] == -1).all()


def test_two_clouds_ragged_other_queries(dev):
    """points1 != points2, ragged lengths, per-cloud radius, queries outside the grid box."""
    g = torch.Generator().manual_seed(5)
    p2 = (torch.rand(3, 900, 3, generator=g) - 0.5) * 2
    p1 = (torch.rand(3, 400, 3, generator=g) - 0.5) * 3.0  # some outside the bbox of p2
    l2 = torch.tensor([900, 17, 500])
    l1 = torch.tensor([400, 5, 123])
    r = torch.tensor([0.3, 0.9, 0.15])
    d, i, nn, _ = _cmp(dev, p1, p2, l1, l2, 5, r)
    assert (i[1, 5:] == -1).all()  # rows beyond lengths1 are empty


def test_k1_insert_shape(dev):
    """insert(): K=1 nearest selected ref point within 4r (levelset_sampling.py:200-202)."""
    g = torch.Generator().manual_seed(8)
    pts = sphere_cloud(3000, seed=3)
    ref = torch.nn.functional.normalize(torch.randn(1, 40, 3, generator=g), dim=-1)
    _cmp(dev, pts, ref, torch.tensor([3000]), None, 1, 0.35)


def test_duplicates_and_ties_lower_index_first(dev):
    base = cube_cloud(200, seed=9)
    p = torch.cat([base, base, base[:, :50]], dim=1)  # exact duplicates -> d2 ties
    _cmp(dev, p, p, None, None, 8, 0.4, same=True)


def test_degenerate_clouds(dev):
    from iso_points_amd import frnn
    one = torch.zeros(1, 1, 3)
    _cmp(dev, one, one, None, None, 3, 0.5, same=True)
    same_pt = torch.ones(1, 50, 3) * 0.25
    _cmp(dev, same_pt, same_pt, None, None, 4, 0.1, same=True)
    empty = torch.zeros(1, 0, 3, device=dev)
    d, i, nn, _ = frnn.frnn_grid_points(empty, empty, K=3, r=0.1, return_nn=True)
    assert d.shape == (1, 0, 3) and i.shape == (1, 0, 3) and nn.shape == (1, 0, 3, 3)


def test_grid_reuse(dev):
    """The opaque grid can be passed back for the same points2 (point_processing.py:73-84)."""
    O = _oracle()
    from iso_points_amd import frnn
    p2 = sphere_cloud(4000, seed=21)
    q = sphere_cloud(1000, seed=22)
    _, _, _, grid = frnn.frnn_grid_points(p2.to(dev), p2.to(dev), K=5, r=0.1)
    d, i, _, _ = frnn.frnn_grid_points(q.to(dev), p2.to(dev), K=5, r=0.1, grid=grid)
    d_ref, i_ref, _, _ = O.frnn_grid_points(q, p2, K=5, r=0.1)
    assert torch.equal(i.cpu(), i_ref) and torch.equal(d.cpu(), d_ref)


def test_100k_against_tree_assisted_oracle(dev):
    """BASELINE.json configs[1] size: 100k points, K=9, tree radius."""
    O = _oracle()
    P = 100000
    p = sphere_cloud(P, seed=100)
    r = O.search_radius(p, torch.tensor([P]), 8)
    _cmp(dev, p, p, None, None, 9, r, same=True)


def test_frnn_gather(dev):
    O = _oracle()
    from iso_points_amd import frnn
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 300, 5, generator=g)
    idx = torch.randint(-1, 300, (2, 100, 6), generator=g)
    out = frnn.frnn_gather(x.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), O.frnn_gather(x, idx))


def test_low_level_2d_grid_api(dev):
    """frnn._C.insert_points_cuda / counting_sort_cuda + prefix_sum_cuda exactly as
    EllipticalRasterizer.backward drives them (DSS/core/rasterizer.py:887-929)."""
    from iso_points_amd import frnn
    from iso_points_amd.prefix_sum import prefix_sum_cuda
    g = torch.Generator().manual_seed(12)
    N, P = 2, 1500
    pts = ((torch.rand(N, P, 2, generator=g) - 0.5) * 1.6).to(dev)
    lengths = torch.tensor([1500, 700], device=dev)
    params = torch.zeros(N, 6, device=dev)
    G = 0
    for i in range(N):
        n = int(lengths[i])
        mn, mx = pts[i, :n].min(0)[0], pts[i, :n].max(0)[0]
        cell = 0.05
        params[i, :2] = mn
        params[i, 2] = 1 / cell
        params[i, 3:5] = torch.floor((mx - mn) / cell) + 1
        params[i, 5] = params[i, 3] * params[i, 4]
        G = max(G, int(params[i, 5]))
    cnt = torch.zeros(N, G, dtype=torch.int32, device=dev)
    cell_id = torch.full((N, P), -1, dtype=torch.int32, device=dev)
    slot = torch.full((N, P), -1, dtype=torch.int32, device=dev)
    frnn._C.insert_points_cuda(pts, lengths, params, cnt, cell_id, slot, G)
    off = torch.zeros(N, G, dtype=torch.int32, device=dev)
    for i in range(N):
        prefix_sum_cuda(cnt[i], params[i, 5].item(), off[i])
    srt = torch.zeros(N, P, 2, device=dev)
    sidx = torch.full((N, P), -1, dtype=torch.int32, device=dev)
    frnn._C.counting_sort_cuda(pts, lengths, cell_id, slot, off, srt, sidx)
    for i in range(N):
        n = int(lengths[i])
        gt = int(params[i, 5])
        gx = torch.floor((pts[i, :n, 0] - params[i, 0]) * params[i, 2]).long()
        gy = torch.floor((pts[i, :n, 1] - params[i, 1]) * params[i, 2]).long()
        c_ref = (gx * int(params[i, 4]) + gy).int()
        assert torch.equal(cell_id[i, :n], c_ref)
        cnt_ref = torch.bincount(c_ref.long(), minlength=gt).int()
        assert torch.equal(cnt[i, :gt], cnt_ref)
        assert torch.equal(off[i, :gt], (torch.cumsum(cnt_ref, 0) - cnt_ref).int())
        perm = sidx[i, :n].long()
        assert torch.equal(torch.sort(perm)[0], torch.arange(n, device=dev))
        assert torch.equal(srt[i, :n], pts[i, perm])
        assert (cell_id[i, perm][1:] >= cell_id[i, perm][:-1]).all()  # cell order
        assert (sidx[i, n:] == -1).all()


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100000, 2146689])
def test_prefix_sum_sizes(dev, n):
    from iso_points_amd.prefix_sum import prefix_sum_cuda
    g = torch.Generator().manual_seed(n)
    cnt = torch.randint(0, 7, (n,), generator=g, dtype=torch.int32).to(dev)
    off = torch.empty_like(cnt)
    prefix_sum_cuda(cnt, n, off)
    ref = torch.cumsum(cnt.long(), 0) - cnt.long()
    assert torch.equal(off.long(), ref)


@pytest.mark.parametrize("K", [7, 20])
def test_tail_kernel_isolated_queries(dev, K):
    """Queries a single lane cannot finish within two rings of cells (isolated outliers next to a
    dense surface, large radius) are served by the wave-per-query tail kernel: same exact result."""
    g = torch.Generator().manual_seed(77)
    dense = sphere_cloud(60000, seed=5)
    outl = torch.nn.functional.normalize(torch.randn(1, 40, 3, generator=g), dim=-1) * \
        (1.0 + 0.05 + 0.1 * torch.rand(1, 40, 1, generator=g))
    far = torch.tensor([[[3.0, 3.0, 3.0], [-2.5, 0.0, 0.0]]])
    p = torch.cat([dense, outl, far], dim=1)
    d, i, nn, grid = _cmp(dev, p, p, None, None, K, 0.2, same=True)
    assert int(grid.tail_counts.sum()) > 0
    # separate query set with points far outside the grid box
    q = torch.cat([outl * 1.5, far + 0.1, dense[:, :100]], dim=1)
    d, i, nn, grid = _cmp(dev, q, p, None, None, K, 0.45)
    assert int(grid.tail_counts.sum()) > 0
