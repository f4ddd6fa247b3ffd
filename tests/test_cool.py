"""`.cool` / `.mcool` ingestion without cooler or h5py (SURVEY.md section 8f row 4;
peakachu/score_genome.py:26-35,55-57).

The fixtures are genuine HDF5 files -- written by h5py 3.3.0 / libhdf5 1.10.6 (this image's
Anaconda interpreter, tools/make_cool_fixture.py) in the layout cooler gives its files: chunked
datasets with gzip 6 + shuffle, an enum-typed bins/chrom, fixed-length names, variable-length
string attributes.  `cool_small_expected.npz` holds what
`cooler.Cooler(path).matrix(balance=..., sparse=True).fetch(chrom)` returns for such a file,
computed by the generator from the same arrays; `h5lite_types.h5` exercises the rest of the
reader (n-D chunks, big-endian, fletcher32, deep and large groups, attributes)."""
import os

import numpy as np
import pytest
from scipy import sparse

from peakachu_amd import cool, h5lite, io, utils

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_h5lite_reads_what_libhdf5_wrote():
    z = np.load(os.path.join(G, "h5lite_types_expected.npz"))
    with h5lite.File(os.path.join(G, "h5lite_types.h5")) as f:
        assert len(z.files) == 82
        for k in z.files:
            d = f[k.replace("|", "/")]
            a, e = np.asarray(d.read()), z[k]
            assert a.shape == e.shape and np.array_equal(a, e), k
            assert a.dtype.kind == e.dtype.kind and a.dtype.itemsize == e.dtype.itemsize, k
        assert len(f["many"].keys()) == 70 and f["a/b/c"].keys() == ["deep"]
        assert f.attrs["float"] == 1.5 and f.attrs["fixed"] == "fixed-length"
        assert f.attrs["text"] == "variable length é" and list(f.attrs["ints"]) == [1, 2, 3]
        assert f["f32"].attrs == {"unit": "m"}
        # ranged reads touch only the chunks they need and agree with the whole
        assert np.array_equal(f["checksummed"][100:1030], z["checksummed"][100:1030])
        assert np.array_equal(f["u64"][3:9], z["u64"][3:9]) and f["u64"][5:5].size == 0
        assert np.array_equal(f["contiguous_f64"][10:20], z["contiguous_f64"][10:20])
        with pytest.raises(KeyError):
            f["nothing/here"]


@pytest.mark.parametrize("uri", ["cool_small.cool", "cool_small.mcool::/resolutions/10000",
                                 "cool_small.mcool::resolutions/10000"])
def test_coolfile_serves_what_cooler_serves(uri):
    z = np.load(os.path.join(G, "cool_small_expected.npz"))
    c = cool.CoolFile(os.path.join(G, uri))
    assert c.chromnames == [str(s) for s in z["chromnames"]] and c.binsize == int(z["binsize"])
    assert [c.chromsizes[k] for k in c.chromnames] == [int(v) for v in z["chromsizes"]]
    for name in c.chromnames:
        n = int(z[name + "/n"])
        assert c.chrom_bins(name) == n == io.chrom_bins(c, name)
        # ("KR" is divisive by its name, as in cooler: (1 / bias_i) * (1 / bias_j) * count)
        for tag, bal in (("raw", False), ("weight", "weight"), ("KR", "KR"), ("weight", True)):
            M = c.matrix(balance=bal, sparse=True).fetch(name)
            assert M.shape == (n, n) and sparse.isspmatrix_coo(M)
            assert M.data.dtype == (np.int32 if bal is False else np.float64)  # counts stay integers, as in cooler
            A = sparse.csr_matrix((M.data.astype(np.float64), (M.row, M.col)), shape=M.shape)
            A.sort_indices()
            assert np.array_equal(A.indptr, z["%s/%s/indptr" % (name, tag)])
            assert np.array_equal(A.indices, z["%s/%s/indices" % (name, tag)])
            assert np.array_equal(A.data.view(np.uint64), z["%s/%s/data" % (name, tag)].view(np.uint64))
            if bal is not False:  # the two weights are multiplied first: exactly symmetric, NaN where a weight is
                assert (abs(A - A.T) > 0).nnz == 0 or np.isnan(A.data).any()
        # the divisive_weights attribute of a column overrides the default its name implies
        for tag, bal in (("KR", "DIV"), ("weight", "VC")):
            M = c.matrix(balance=bal, sparse=True).fetch(name)
            A = sparse.csr_matrix((M.data, (M.row, M.col)), shape=M.shape)
            A.sort_indices()
            assert np.array_equal(A.data.view(np.uint64), z["%s/%s/data" % (name, tag)].view(np.uint64))
        for col in ("weight", "KR"):
            assert np.array_equal(c.bins().fetch(name)[col].values, z[name + "/" + col], equal_nan=True)
    with pytest.raises(ValueError, match="Unknown sequence label"):
        c.matrix(balance=False, sparse=True).fetch("chr9")
    with pytest.raises(ValueError, match="No column 'bins/ICE'"):
        c.matrix(balance="ICE", sparse=True).fetch("chr1")
    c.close()


def test_open_map_and_errors(tmp_path):
    lib = io.open_map(os.path.join(G, "cool_small.cool"))
    assert type(lib).__name__ in ("CoolFile", "Cooler") and lib.chromnames[0] == "chr1"
    with pytest.raises(ValueError, match="resolutions/<binsize>"):
        cool.CoolFile(os.path.join(G, "cool_small.mcool"))
    with pytest.raises(KeyError):
        cool.CoolFile(os.path.join(G, "cool_small.mcool") + "::/resolutions/5000")
    bad = tmp_path / "not.cool"
    bad.write_bytes(b"PK\x03\x04" + b"\x00" * 4000)
    with pytest.raises(h5lite.H5FormatError, match="not an HDF5 file"):
        cool.CoolFile(str(bad))
    assert cool.is_cool(os.path.join(G, "cool_small.mcool") + "::/resolutions/10000") and not cool.is_cool(str(bad))


def test_libver_latest_files():
    """Files written with libver='latest' (version-2 object headers, compact links, dense
    attribute storage, fixed-array and extensible-array chunk indexes): the cooler reads like
    the default-format one (its root attributes sit in a fractal heap and are skipped: the bin
    size then comes from bins/start, bins/end), and deep indexes -- super blocks, paged data
    blocks, never-written chunks -- give the data the genuine library stored."""
    z = np.load(os.path.join(G, "cool_small_expected.npz"))
    c = cool.CoolFile(os.path.join(G, "cool_small_latest.cool"))
    assert c.binsize == 10000 and c.chromnames == ["chr1", "chr2", "chrX"]
    for name in c.chromnames:
        for tag, bal in (("raw", False), ("weight", "weight")):
            M = c.matrix(balance=bal, sparse=True).fetch(name)
            A = sparse.csr_matrix((M.data.astype(np.float64), (M.row, M.col)), shape=M.shape)
            A.sort_indices()
            assert np.array_equal(A.indices, z["%s/%s/indices" % (name, tag)])
            assert np.array_equal(A.data.view(np.uint64), z["%s/%s/data" % (name, tag)].view(np.uint64))
    c.close()

    def ramp(n):
        return (np.arange(n, dtype=np.int64) * 7919 % 30011).astype(np.int16)
    with h5lite.File(os.path.join(G, "h5lite_latest_many.h5")) as f:
        assert f.keys() == ["ea_20000_chunks", "ea_3000_chunks", "ea_filtered_5000_chunks", "ea_sparse",
                            "fa_5000_chunks", "fa_filtered_3000_chunks"]
        for k in f.keys():
            a = f[k].read()
            if k == "ea_sparse":
                e = np.zeros(64 * 300, np.int16)
                e[64 * 250:64 * 251] = ramp(64)
            else:
                e = ramp(a.size)
                assert np.array_equal(f[k][1001:7777], e[1001:7777]), k
            assert a.dtype == np.int16 and np.array_equal(a, e), k


def test_fetch_hands_utils_tocsr_a_ready_made_csr():
    """The matrix a fetch returns is a coo_matrix in canonical order with its row pointer attached;
    utils.tocsr (the reference's next step, peakachu/utils.py:10-15) assembles the CSR from the
    parts -- the same CSR the plain COO -> CSR conversion makes -- unless the arrays were replaced."""
    from peakachu_amd import utils
    c = cool.CoolFile(os.path.join(G, "cool_small.cool"))
    for bal in (False, "weight", "KR"):
        X = c.matrix(balance=bal, sparse=True).fetch("chr1")
        assert sparse.isspmatrix_coo(X) and X._pk_csr_parts[1] is X.col
        key = np.lexsort((X.col, X.row))
        assert np.array_equal(key, np.arange(key.size))  # canonical order
        fast = utils.tocsr(X)
        assert fast.has_canonical_format and fast.data.dtype == np.float64
        slow = sparse.csr_matrix((X.data, (X.row, X.col)), shape=X.shape, dtype=float)
        slow.sum_duplicates()
        slow.sort_indices()
        for k in ("indptr", "indices"):
            assert np.array_equal(getattr(fast, k), getattr(slow, k))
        assert np.array_equal(fast.data.view(np.uint64), slow.data.view(np.uint64))
        X.data = X.data * 2.0  # a caller who edits the COO gets the plain conversion
        assert np.array_equal(utils.tocsr(X).data.view(np.uint64), (slow.data * 2.0).view(np.uint64))
    c.close()


def test_mirror_by_scatter_equals_the_transposed_sum():
    """Pixel lists with explicit zero counts take the numpy path (scipy's sorted-row addition
    would drop the zeros): same canonical order, zeros kept."""
    rng = np.random.default_rng(5)
    n = 300
    i = np.sort(rng.integers(0, n, 4000)).astype(np.int32)
    j = np.minimum(i + rng.integers(0, 40, i.size), n - 1).astype(np.int32)
    key = np.unique(i.astype(np.int64) * n + j)
    i, j = (key // n).astype(np.int32), (key % n).astype(np.int32)
    v = rng.integers(0, 5, i.size).astype(np.int32)  # zeros among them
    row, col, val, indptr = utils._mirror_by_scatter(i, j, v, n)
    off = i != j
    R, Cc, V = np.r_[i, j[off]], np.r_[j, i[off]], np.r_[v, v[off]]
    o = np.lexsort((Cc, R))
    assert np.array_equal(row, R[o]) and np.array_equal(col, Cc[o]) and np.array_equal(val, V[o])
    assert np.array_equal(indptr, np.searchsorted(R[o], np.arange(n + 1)))


def test_row_pointer_is_only_trusted_for_canonical_untouched_matrices():
    """utils.tocsr assembles the CSR from the reader's row pointer only while the COO is the object
    the reader made and its entries are in canonical order; anything else takes the reference's
    conversion (peakachu/utils.py:10-15), which sorts and sums duplicates."""
    from peakachu_amd import utils
    assert utils.is_canonical(np.array([0, 0, 1, 2, 2]), np.array([1, 4, 0, 2, 3]))
    assert not utils.is_canonical(np.array([0, 0, 1]), np.array([4, 1, 0]))      # columns out of order
    assert not utils.is_canonical(np.array([0, 0, 1]), np.array([2, 2, 0]))      # a duplicate
    assert not utils.is_canonical(np.array([1, 0, 2]), np.array([0, 1, 2]))      # rows out of order
    row = np.array([0, 0, 1, 2], np.int32); col = np.array([0, 2, 1, 2], np.int32)
    data = np.array([1.0, 2.0, 3.0, 4.0]); indptr = np.array([0, 2, 3, 4], np.int32)
    X = cool.sparse_coo(data, row, col, 3, indptr)
    A = utils.tocsr(X)
    assert A.indices is X.col or np.shares_memory(A.indices, X.col)       # the fast path
    want = sparse.csr_matrix((data, (row, col)), shape=(3, 3)).toarray()
    assert np.array_equal(A.toarray(), want)
    # the same object with its row array replaced (or another shape): the parts no longer describe it
    Y = cool.sparse_coo(data, row, col, 3, indptr)
    Y.row = np.array([2, 0, 1, 0], np.int32)
    B = utils.tocsr(Y)
    assert not np.shares_memory(B.indices, Y.col)
    assert np.array_equal(B.toarray(), sparse.csr_matrix((data, (Y.row, col)), shape=(3, 3)).toarray())
    Z = cool.sparse_coo(data, row, col, 3, indptr)
    Z._shape = (4, 4)
    assert utils.tocsr(Z).shape == (4, 4) and not np.shares_memory(utils.tocsr(Z).indices, Z.col)


def test_chromosomes_read_on_two_threads_and_shared_arrays(monkeypatch):
    """score_genome reads the next two chromosomes on background threads (positional reads, a
    locked pixel cache): what they deliver equals what one thread reads one after the other.
    The matrices of one chromosome share its cached index arrays, which are read-only.
    (PK_UPPER=0: the host matrices of the reference's flow; the default hands the device the
    pixel table as it is, see test_upper_pixels_equal_the_mirrored_matrices.)"""
    from peakachu_amd import score_genome
    monkeypatch.setenv("PK_UPPER", "0")
    path = os.path.join(G, "cool_small.cool")
    ref = cool.CoolFile(path)
    names = ref.chromnames * 3
    want = [score_genome.fetch_inputs(ref, k, "weight") for k in names]
    for depth in ("1", "2", "3"):
        os.environ["PK_PREFETCH"] = depth
        try:
            c = cool.CoolFile(path)
            got = list(score_genome.prefetched(c, names, "weight"))
        finally:
            del os.environ["PK_PREFETCH"]
        assert [k for k, _ in got] == names
        # every chromosome's pixels were read ONCE per visit, whatever the number of readers
        # (round 3: a cache of two entries under three threads lost the entry between a
        # chromosome's balanced and raw fetch)
        assert c.pixel_reads <= len(names), (depth, c.pixel_reads)
        for (_, (M, R, wts)), (M0, R0, w0) in zip(got, want):
            for A, B in ((M, M0), (R, R0)):
                assert np.array_equal(A.row, B.row) and np.array_equal(A.col, B.col)
                assert np.array_equal(A.data, B.data, equal_nan=True) and A.data.dtype == B.data.dtype
            assert np.array_equal(wts, w0, equal_nan=True)
            assert M.row is R.row or np.shares_memory(M.row, R.row)
            with pytest.raises(ValueError, match="read-only"):
                R.data[0] = 1
            M.data[0] = M.data[0]  # the balanced values are the caller's own
        c.close()
    ref.close()


@pytest.mark.parametrize("uri", ["cool_small.cool", "cool_small_latest.cool"])
def test_upper_pixels_equal_the_mirrored_matrices(uri):
    """What score_genome hands the device by default -- the chromosome as the file stores it
    (CoolFile.upper: upper triangle incl. the trans pixels that share its rows, CoolFile.bias) --
    mirrors on the host (UpperPixels.symmetric, the on-demand copy behind Chromosome.M / raw_M)
    into exactly the matrices of the reference's flow: matrix(balance=...).fetch -> utils.tocsr."""
    from peakachu_amd import score_genome
    c = cool.CoolFile(os.path.join(G, uri))
    saw_trans = False
    for name in c.chromnames:
        px = c.upper(name)
        assert px.cols.dtype == np.int32 and px.indptr[0] == 0 and px.indptr[-1] == px.cols.size
        saw_trans |= bool((px.cols >= px.n).any())
        raw = utils.tocsr(c.matrix(balance=False, sparse=True).fetch(name))
        got = px.symmetric()
        assert np.array_equal(got.indptr, raw.indptr) and np.array_equal(got.indices, raw.indices)
        assert np.array_equal(got.data, raw.data)
        for col in ("weight", "KR", "DIV", "VC"):
            bias, column = c.bias(col, name)
            assert np.array_equal(column, c.bins().fetch(name)[col].values, equal_nan=True)
            bal = utils.tocsr(c.matrix(balance=col, sparse=True).fetch(name))
            got = px.symmetric(bias)
            assert np.array_equal(got.indices, bal.indices)
            assert np.array_equal(got.data.view(np.uint64), bal.data.view(np.uint64))
        inp = score_genome.fetch_inputs(c, name, "weight")
        assert isinstance(inp, score_genome.UpperInputs) and inp.pixels.n == px.n
    assert saw_trans  # the fixture holds trans pixels: they ride along and are not part of the matrix
    c.close()


def test_upper_pixels_of_a_non_conforming_table_are_sorted_and_summed():
    """A pixel table with unsorted columns and a duplicate: the host copy equals the reference's
    conversion (COO -> CSR sums duplicates, peakachu/utils.py:10-15)."""
    indptr = np.array([0, 3, 4, 4], np.int32)
    px = utils.UpperPixels(3, indptr, np.array([2, 0, 2, 1], np.int32), np.array([5, 1, 7, 3], np.int32))
    M = px.symmetric().toarray()
    assert np.array_equal(M, np.array([[1., 0., 12.], [0., 3., 0.], [12., 0., 0.]]))


def test_chunk_pipeline_in_c_equals_the_python_one(monkeypatch):
    """A ranged read runs its chunks through pk_host_unfilter_chunks (inflate + un-shuffle on host
    threads, include/peakachu_hip.h); without the library h5lite does the same in Python.  Same
    arrays, whatever the range's position inside the chunks."""
    for name in ("cool_small.cool", "cool_small_latest.cool", "h5lite_types.h5", "h5lite_latest_many.h5"):
        f = h5lite.File(os.path.join(G, name))
        sets = []

        def walk(g):
            for k in g.keys():
                o = g[k]
                if isinstance(o, h5lite.Group):
                    walk(o)
                elif len(o.shape) == 1 and o.shape[0] > 3 and o._type.kind not in ("vlen", "string"):
                    sets.append(o)
        walk(f)
        assert sets
        rng = np.random.default_rng(3)
        for d in sets:
            n = d.shape[0]
            for _ in range(4):
                lo = int(rng.integers(0, n - 1))
                hi = int(rng.integers(lo + 1, n + 1))
                assert h5lite._native_unfilter() is not None
                fast = d[lo:hi]
                monkeypatch.setattr(h5lite, "_NATIVE", None)
                slow = d[lo:hi]
                monkeypatch.undo()
                assert fast.dtype == slow.dtype and np.array_equal(fast.view(np.uint8), slow.view(np.uint8)), (name, d.name)
        f.close()
