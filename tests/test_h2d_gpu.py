"""bd_h2d_* (csrc/h2d.hip): the host batch -> fp32 device tensor leg of data_to_input (layers/common/pre_processing.py:13).  Transport only:
the device tensor must equal numpy's own astype(float32) of the host array bit for bit, for every source dtype, chunking and thread count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_h2d_matches_numpy_astype(threads):
    from basedet_amd import ops
    st = ops.HostStager("cuda:0", threads)
    assert st.threads >= 1 and (threads == 0 or st.threads == threads)
    rng = np.random.default_rng(threads)
    cases = [rng.random((2, 3, 37, 53)), rng.random((1, 3, 800, 1344)), (rng.random((3, 3, 64, 96)) * 255).astype(np.float32),
             rng.integers(0, 256, (2, 3, 50, 70), dtype=np.uint8), rng.standard_normal((5,)) * 1e30, np.zeros((0, 3, 4, 4))]
    for arr in cases:
        for chunk in (0, 1000, 1 << 22):
            got = st.submit(arr, chunk_elems=chunk)
            torch.cuda.synchronize()
            want = arr.astype(np.float32)
            assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want, equal_nan=True), (arr.dtype, arr.shape, chunk)
    # back-to-back submits reuse the staging buffer: the second must wait for the first one's copies
    a, b = rng.random((4, 3, 256, 256)), rng.random((4, 3, 256, 256))
    da = st.submit(a)
    db = st.submit(b)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy(), a.astype(np.float32)) and np.array_equal(db.cpu().numpy(), b.astype(np.float32))
    st.close()


def test_model_preprocess_takes_host_batches_of_every_dtype():
    """FPNDetector.pre_process on float64 / float32 / uint8 host batches and on a device tensor: identical padded-normalised input."""
    from basedet_amd.models import RetinaNet
    from tests.test_model_gpu import _setup
    cfg, params, batch = _setup("resnet18", 2, (100, 130))
    model = RetinaNet(cfg, params=params)
    u8 = np.random.default_rng(0).integers(0, 256, batch["data"].shape, dtype=np.uint8)
    ref = None
    for data in (u8, u8.astype(np.float64), u8.astype(np.float32), torch.from_numpy(u8.astype(np.float32)).cuda(),
                 torch.from_numpy(u8.astype(np.float64))):
        pre = model.pre_process(dict(batch, data=data))
        torch.cuda.synchronize()
        x = pre["plan"].x_halo.clone()
        if ref is None:
            ref = x
        assert torch.equal(x, ref)
    assert float(ref.float().abs().sum()) > 0


def test_h2d_pool_survives_many_submits_of_changing_sizes():
    """The worker pool is woken once per submit (generation counter + condition variable): 120 back-to-back submits of changing sizes,
    dtypes and chunkings, each checked -- a lost wake-up would hang (pytest-timeout), a stale job descriptor would copy the wrong range."""
    from basedet_amd import ops
    st = ops.HostStager("cuda:0", 8)
    rng = np.random.default_rng(5)
    pending = []
    for it in range(120):
        n = int(rng.integers(1, 400_000))
        arr = (rng.random(n) * 255) if it % 3 else rng.integers(0, 256, n, dtype=np.uint8)
        if it % 3 == 2:
            arr = arr.astype(np.float32)
        out = torch.empty((n,), dtype=torch.float32, device="cuda")
        st.submit(arr, out, chunk_elems=int(rng.choice([0, 777, 4096, 100_000])))
        pending.append((arr, out))
        if len(pending) == 8:
            torch.cuda.synchronize()
            for a, o in pending:
                assert np.array_equal(o.cpu().numpy(), a.astype(np.float32))
            pending = []
    torch.cuda.synchronize()
    for a, o in pending:
        assert np.array_equal(o.cpu().numpy(), a.astype(np.float32))
    st.close()
