"""Dataset readers + batch feed on synthetic files written in the reference's on-disk formats
(image_datasets.py:183-470).  The reference module itself cannot be imported in the build container (blobfile, mpi4py and
torchvision are absent), so these tests pin the readers to the FORMAT: hand-assembled bytes, file-name labels, the
documented normalisations and the `[shard:][::num_shards]` stride."""
import gzip
import io
import os
import struct

import numpy as np
import pytest
import torch as th

from causaldiffae_amd import image_datasets as ds

PIL = pytest.importorskip("PIL.Image")


def test_idx_known_bytes(tmp_path):
    # 2x3 u8 matrix, header assembled by hand: 00 00 08 02 | 00000002 | 00000003 | payload
    raw = bytes([0, 0, 8, 2]) + struct.pack(">II", 2, 3) + bytes([1, 2, 3, 250, 251, 252])
    p = tmp_path / "a-idx2-ubyte"
    p.write_bytes(raw)
    a = ds.read_idx(p)
    assert a.dtype == np.uint8 and a.tolist() == [[1, 2, 3], [250, 251, 252]]
    pz = tmp_path / "a-idx2-ubyte.gz"
    with gzip.open(pz, "wb") as f:
        f.write(raw)
    assert np.array_equal(ds.read_idx(pz), a)
    # big-endian int32 vector
    p2 = tmp_path / "b-idx1-int"
    p2.write_bytes(bytes([0, 0, 0x0C, 1]) + struct.pack(">I", 2) + struct.pack(">ii", -5, 70000))
    assert ds.read_idx(p2).tolist() == [-5, 70000]
    with pytest.raises(ValueError):
        bad = tmp_path / "bad"
        bad.write_bytes(raw[:-1])
        ds.read_idx(bad)


@pytest.mark.parametrize("dtype", [np.uint8, np.int32, np.float32])
def test_idx_roundtrip(tmp_path, dtype):
    rng = np.random.RandomState(0)
    a = (rng.rand(5, 4, 3) * 100).astype(dtype)
    ds.write_idx(a, tmp_path / "x.gz")
    b = ds.read_idx(tmp_path / "x.gz")
    assert b.dtype == np.dtype(dtype) and np.array_equal(a, b)


def _make_morpho(root, n_train=23, n_test=40):
    import pandas as pd
    os.makedirs(root, exist_ok=True)
    rng = np.random.RandomState(1)
    out = {}
    for prefix, n in (("train", n_train), ("t10k", n_test)):
        imgs = rng.randint(0, 256, size=(n, 28, 28)).astype(np.uint8)
        labs = rng.randint(0, 10, size=(n,)).astype(np.uint8)
        df = pd.DataFrame({"thickness": rng.rand(n) * 4 + 1, "intensity": rng.rand(n) * 200 + 50, "slant": rng.randn(n)})
        ds.write_idx(imgs, os.path.join(root, prefix + "-images-idx3-ubyte.gz"))
        ds.write_idx(labs, os.path.join(root, prefix + "-labels-idx1-ubyte.gz"))
        df.to_csv(os.path.join(root, prefix + "-morpho.csv"), index_label="index")
        out[prefix] = (imgs, labs, df)
    return out


def test_morphomnist_reader_and_shards(tmp_path):
    root = str(tmp_path / "morphomnist")
    truth = _make_morpho(root)
    imgs, labs, df = truth["train"]
    full = ds.read_morphomnist(root, "train")
    assert full.images.shape == (23, 28, 28, 1) and np.array_equal(full.images[..., 0], imgs)
    assert full.cond["y"].dtype == np.int64 and np.array_equal(full.cond["y"], labs)
    want_c = np.stack([df["thickness"], df["intensity"]], 1).astype(np.float32)      # RAW metrics, thickness first
    assert np.array_equal(full.cond["c"], want_c)
    for shard in range(3):
        p = ds.read_morphomnist(root, "train", shard, 3)
        assert np.array_equal(p.images[..., 0], imgs[shard::3]) and np.array_equal(p.cond["c"], want_c[shard::3])
    test = ds.read_morphomnist(root, "test")
    assert len(test) == 40 and np.array_equal(test.images[..., 0], truth["t10k"][0])
    val = ds.read_morphomnist(root, "val")
    perm = th.randperm(40, generator=th.Generator().manual_seed(42)).numpy()
    assert len(val) == 4 and np.array_equal(val.images[..., 0], truth["t10k"][0][perm[36:]])


def _make_pendulum(root, split, n=10):
    d = os.path.join(root, split)
    os.makedirs(d, exist_ok=True)
    rng = np.random.RandomState(2)
    items = {}
    for i in range(n):
        lab = (int(rng.randint(-40, 44)), int(rng.randint(60, 148)), int(rng.randint(3, 12)), int(rng.randint(3, 19)))
        if ("a_%d_%d_%d_%d.png" % lab) in items:
            continue
        arr = rng.randint(0, 256, size=(96, 96, 4)).astype(np.uint8)
        PIL.fromarray(arr, "RGBA").save(os.path.join(d, "a_%d_%d_%d_%d.png" % lab))
        items["a_%d_%d_%d_%d.png" % lab] = (lab, arr)
    return items


def test_pendulum_reader(tmp_path):
    root = str(tmp_path / "pendulum")
    items = _make_pendulum(root, "train")
    names = sorted(items)
    pool = ds.read_pendulum(root, "train")
    assert pool.images.shape == (len(names), 96, 96, 4)
    scale = np.array([[2, 42], [104, 44], [7.5, 4.5], [11, 8]], np.float32)
    for i, n in enumerate(names):
        lab, arr = items[n]
        assert np.array_equal(pool.images[i], arr)
        want = (np.asarray(lab, np.float32) - scale[:, 0]) / scale[:, 1]
        assert np.array_equal(pool.cond["c"][i], want)
    # negative first label parses (names look like a_-12_100_5_9.png)
    assert any(items[n][0][0] < 0 for n in names)
    p1 = ds.read_pendulum(root, "train", 1, 2)
    assert np.array_equal(p1.images, pool.images[1::2]) and np.array_equal(p1.cond["c"], pool.cond["c"][1::2])
    x, cond = next(ds.Feed(pool, 4, shuffle=False))
    assert x.shape == (4, 4, 96, 96) and x.dtype == th.float32
    assert th.equal(x[2], th.from_numpy(pool.images[2].astype(np.float32) / 255.0).permute(2, 0, 1))     # ToTensor
    assert cond["c"].shape == (4, 4) and cond["c"].dtype == th.float32


def _png_bytes(arr):
    b = io.BytesIO()
    PIL.fromarray(arr, "RGB").save(b, format="PNG")
    return b.getvalue()


def test_circuit_reader(tmp_path):
    root = str(tmp_path / "causal_circuit")
    os.makedirs(root)
    rng = np.random.RandomState(3)
    all_lat, all_img = [], []
    for k in range(5):
        n = 3
        imgs = np.empty((n, 2), dtype=object)
        arrs = rng.randint(0, 256, size=(n, 2, 256, 256, 3)).astype(np.uint8)
        for i in range(n):
            for j in range(2):
                imgs[i, j] = _png_bytes(arrs[i, j])
        lat = rng.rand(n, 2, 4).astype(np.float32)
        np.savez(os.path.join(root, f"train-{k}.npz"), imgs=imgs, original_latents=lat)
        all_lat.append(lat[:, 0])
        all_img.append(arrs[:, 0])
    np.savez(os.path.join(root, "test.npz"), imgs=imgs, original_latents=lat)
    lat, arrs = np.concatenate(all_lat), np.concatenate(all_img)
    pool = ds.read_circuit(root, "train")
    assert pool.images.shape == (15, 128, 128, 3)
    assert np.array_equal(pool.cond["c"], lat[:, [3, 2, 1, 0]])
    want = np.asarray(PIL.fromarray(arrs[7], "RGB").resize((128, 128), PIL.BILINEAR))
    assert np.array_equal(pool.images[7], want)
    p = ds.read_circuit(root, "train", 1, 4)
    assert np.array_equal(p.images, pool.images[1::4]) and np.array_equal(p.cond["c"], pool.cond["c"][1::4])
    assert len(ds.read_circuit(root, "test")) == 3
    # constant image stays constant under the resize and maps to v/255
    g = next(ds.load_data(data_dir=root, batch_size=5, image_size=128, split="train"))
    assert g[0].shape == (5, 3, 128, 128) and th.equal(g[1]["c"], th.from_numpy(pool.cond["c"][:5]))       # not shuffled


def test_image_folder_reader(tmp_path):
    root = tmp_path / "celeba"
    (root / "sub").mkdir(parents=True)
    rng = np.random.RandomState(4)
    a = rng.randint(0, 256, size=(70, 100, 3)).astype(np.uint8)
    b = np.full((64, 64, 3), 200, np.uint8)
    PIL.fromarray(a).save(root / "cat_1.png")
    PIL.fromarray(b).save(root / "sub" / "dog_7.png")
    (root / "notes.txt").write_text("x")
    pool = ds.read_image_folder(str(root), 32, class_cond=True)
    assert pool.images.shape == (2, 32, 32, 3) and pool.cond["y"].tolist() == [0, 1]
    assert np.all(pool.images[1] == 200)
    x, cond = next(ds.Feed(pool, 2, shuffle=False))
    assert abs(float(x[1].mean()) - (200 / 127.5 - 1)) < 1e-6 and x.min() >= -1 and x.max() <= 1
    # 70x100 -> BOX halve (35x50) -> BICUBIC short side 32 (32x46) -> centre crop
    im = PIL.fromarray(a).resize((50, 35), PIL.BOX).resize((46, 32), PIL.BICUBIC)
    assert np.array_equal(pool.images[0], np.asarray(im)[:, 7:39])


def test_feed_epoch_rule():
    imgs = np.arange(10, dtype=np.uint8).reshape(10, 1, 1, 1).repeat(4, axis=3)
    pool = ds.Pool(imgs, {"c": np.arange(10, dtype=np.float32)[:, None]}, div=1.0, shift=0.0)
    f = ds.Feed(pool, 4, shuffle=True, seed=5)
    e = [next(f)[1]["c"][:, 0].tolist() for _ in range(4)]
    assert len(set(e[0] + e[1])) == 8 and len(set(e[2] + e[3])) == 8          # 2 full batches per epoch, tail of 2 dropped
    f2 = ds.Feed(pool, 4, shuffle=True, seed=5)
    assert [next(f2)[1]["c"][:, 0].tolist() for _ in range(4)] == e            # seeded
    g = ds.Feed(pool, 4, shuffle=False)
    assert next(g)[1]["c"][:, 0].tolist() == [0, 1, 2, 3] and next(g)[1]["c"][:, 0].tolist() == [4, 5, 6, 7]
    assert next(g)[1]["c"][:, 0].tolist() == [0, 1, 2, 3]
    x, _ = next(g)
    assert x.shape == (4, 4, 1, 1) and x[:, 0, 0, 0].tolist() == [4, 5, 6, 7]
    with pytest.raises(ValueError):
        ds.Feed(pool, 11)


def test_load_data_dispatch(tmp_path):
    root = str(tmp_path / "morphomnist")
    _make_morpho(root)
    x, cond = next(ds.load_data(data_dir=root, batch_size=8, image_size=28, split="train"))
    assert x.shape == (8, 1, 28, 28) and cond["y"].dtype == th.int64 and cond["c"].shape == (8, 2)
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    with pytest.raises(ValueError):
        next(ds.load_data(data_dir=str(tmp_path / "unknown"), batch_size=2, image_size=8))
    x, cond = next(ds.load_data(data_dir="synthetic", batch_size=2, image_size=8, in_channels=4, class_cond=True))
    assert x.shape == (2, 4, 8, 8) and set(cond) == {"c", "y"}
