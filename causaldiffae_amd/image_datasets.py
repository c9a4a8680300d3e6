"""Data feed: the reference's `load_data` generator (image_datasets.py:69-126) and its four dataset
readers, re-designed for a 288 GB HBM device.

The reference decodes and normalises every item on the host inside a `DataLoader` worker and ships fp32 batches over
PCIe each step.  Here a dataset shard is decoded ONCE into a u8 HWC *pool* (MorphoMNIST train: 47 MB, Pendulum: 258 MB,
CausalCircuit at 128 px: 1.7 GB), the pool is made resident in HBM, and a batch is one `cdae_gather_u8` launch
(index gather + u8 -> fp32 `x/div+shift`, output already in the NHWC storage the UNet consumes).  Labels ride along as small
device tensors.  With `device=None` (or a CPU device) the same feed yields host tensors the way the reference does —
that is the form the CPU tests pin — and the TrainLoop moves them itself.

Rank sharding is the reference's `[shard:][::num_shards]` stride.  File formats:
  * MorphoMNIST-like: `{train,t10k}-images-idx3-ubyte.gz`, `-labels-idx1-ubyte.gz`, `-morpho.csv` (image_datasets.py:183-218)
  * Pendulum: `<root>/<split>/a_<i>_<j>_<k>_<l>.png`, RGBA 96x96, labels in the file name (image_datasets.py:337-377)
  * CausalCircuit: `<root>/{train-0..4,test}.npz` with `imgs[:,0]` PNG bytes and `original_latents[:,0,:]` (image_datasets.py:395-470)
  * image folder ("celeba"): recursive jpg/png listing, BOX halving + BICUBIC + centre crop, [-1,1] (image_datasets.py:128-177)
"""
import gzip
import io
import os
import struct

import numpy as np
import torch as th

_IDX_DTYPES = {0x08: np.uint8, 0x09: np.int8, 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}
_IDX_CODES = {np.dtype(np.uint8): 0x08, np.dtype(np.int8): 0x09, np.dtype(np.int16): 0x0B, np.dtype(np.int32): 0x0C,
              np.dtype(np.float32): 0x0D, np.dtype(np.float64): 0x0E}


# ---------------------------------------------------------------------------------------------------- idx files
def read_idx(path):
    """Parse an (optionally gzipped) IDX file: 2 zero bytes, dtype code, ndim, big-endian u32 dims, row-major payload."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    if len(raw) < 4 or raw[0] != 0 or raw[1] != 0 or raw[2] not in _IDX_DTYPES:
        raise ValueError(f"{path}: not an IDX file")
    nd = raw[3]
    dims = struct.unpack(">" + "I" * nd, raw[4:4 + 4 * nd])
    dt = np.dtype(_IDX_DTYPES[raw[2]])
    n = int(np.prod(dims)) if nd else 1
    body = raw[4 + 4 * nd:]
    if len(body) != n * dt.itemsize:
        raise ValueError(f"{path}: payload is {len(body)} bytes, header promises {n * dt.itemsize}")
    return np.frombuffer(body, dtype=dt).reshape(dims).astype(dt.newbyteorder("="))


def write_idx(arr, path):
    """Inverse of read_idx (used by the tests and by anyone building a MorphoMNIST-like set)."""
    arr = np.ascontiguousarray(arr)
    code = _IDX_CODES[arr.dtype]
    head = bytes([0, 0, code, arr.ndim]) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(head + arr.astype(arr.dtype.newbyteorder(">")).tobytes())


# ---------------------------------------------------------------------------------------------------- pools
class Pool:
    """A decoded dataset shard: `images` u8 [N,H,W,C]; item = images/div+shift (fp32 IEEE division like ToTensor, logical CHW); `cond` maps the
    reference's out_dict keys to per-item arrays."""

    def __init__(self, images, cond, div=255.0, shift=0.0, name=""):
        assert images.dtype == np.uint8 and images.ndim == 4
        self.images, self.cond, self.div, self.shift, self.name = images, cond, float(div), float(shift), name
        for k, v in cond.items():
            assert len(v) == len(images), k

    def __len__(self):
        return len(self.images)

    def subset(self, index):
        return Pool(self.images[index], {k: v[index] for k, v in self.cond.items()}, self.div, self.shift, self.name)


def _stride(a, shard, num_shards):
    return a[shard:][::num_shards]


def read_morphomnist(root, split="train", shard=0, num_shards=1, columns=("thickness", "intensity")):
    """MorphoMNISTLike + get_dataloader_morphomnist's split rule (image_datasets.py:241-333).  x = u8/255 as [1,28,28];
    y = digit (int64); c = RAW [thickness, intensity] (the reference computes a scaled copy and does not use it)."""
    import pandas as pd
    assert split in ("train", "val", "test")
    prefix = "train" if split == "train" else "t10k"
    images = read_idx(os.path.join(root, prefix + "-images-idx3-ubyte.gz"))
    labels = read_idx(os.path.join(root, prefix + "-labels-idx1-ubyte.gz"))
    metrics = pd.read_csv(os.path.join(root, prefix + "-morpho.csv"), usecols=["index"] + list(columns), index_col="index")
    images = _stride(images, shard, num_shards)
    labels = _stride(labels, shard, num_shards)
    c = np.stack([_stride(np.asarray(metrics[col]), shard, num_shards) for col in columns], axis=1).astype(np.float32)
    pool = Pool(np.ascontiguousarray(images[..., None]), {"y": labels.astype(np.int64), "c": c}, name="morphomnist")
    if split == "val":      # random_split(dataset, [0.9 n, 0.1 n], manual_seed(42))[1]
        n = len(pool)
        n_train, n_val = int(n * 0.9), int(n * 0.1)
        perm = th.randperm(n_train + n_val, generator=th.Generator().manual_seed(42)).numpy()
        pool = pool.subset(perm[n_train:n_train + n_val])
    return pool


PENDULUM_SCALE = np.array([[2, 42], [104, 44], [7.5, 4.5], [11, 8]], dtype=np.float64)     # image_datasets.py:353


def read_pendulum(root, split="train", shard=0, num_shards=1):
    """SyntheticLabeled (image_datasets.py:337-377): c = (label - scale[:,0]) / scale[:,1] from the file name; x = RGBA/255.
    Files are taken in sorted order (the reference uses os.listdir order, which is filesystem-dependent)."""
    from PIL import Image
    assert split in ("train", "val", "test")
    d = os.path.join(root, split)
    names = sorted(n for n in os.listdir(d) if n.lower().endswith(".png"))
    labels = np.asarray([list(map(int, n[:-4].split("_")[1:])) for n in names], dtype=np.float64).reshape(len(names), -1)
    names = _stride(names, shard, num_shards)
    labels = _stride(labels, shard, num_shards)
    imgs = []
    for n in names:
        with Image.open(os.path.join(d, n)) as im:
            a = np.asarray(im)
        imgs.append(a[..., None] if a.ndim == 2 else a)
    images = np.stack(imgs).astype(np.uint8) if imgs else np.zeros((0, 96, 96, 4), np.uint8)
    # the reference normalises in fp32 torch scalars: (label - lo) / hi with label an int64 tensor element
    lo, hi = PENDULUM_SCALE[:, 0].astype(np.float32), PENDULUM_SCALE[:, 1].astype(np.float32)
    c = (labels[:, :4].astype(np.float32) - lo) / hi if len(labels) else np.zeros((0, 4), np.float32)
    return Pool(images, {"c": c}, name="pendulum")


def _resize_short_side(im, size):
    from PIL import Image
    w, h = im.size
    if min(w, h) == size:
        return im
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    return im.resize((nw, nh), Image.BILINEAR)


def read_circuit(root, split="train", shard=0, num_shards=1, resolution=128):
    """CausalCircuit (image_datasets.py:395-470): train = train-0..4.npz concatenated, test = test.npz; the image is the
    first of each pair, short side resized to 128 (bilinear), /255; c = original_latents[:,0,[3,2,1,0]].  The reference
    hard-codes '../datasets/causal_circuit/'; here the files are looked up under `root`."""
    from PIL import Image
    assert split in ("train", "val", "test")
    files = [f"train-{k}.npz" for k in range(5)] if split == "train" else ["test.npz"]
    blobs, labels = [], []
    for fn in files:
        with np.load(os.path.join(root, fn), allow_pickle=True) as data:
            labels.append(np.asarray(data["original_latents"])[:, 0, :])
            blobs.extend(list(data["imgs"][:, 0]))
    labels = _stride(np.concatenate(labels, axis=0), shard, num_shards)
    blobs = _stride(blobs, shard, num_shards)
    imgs = []
    for b in blobs:
        with Image.open(io.BytesIO(bytes(b))) as im:
            a = np.asarray(_resize_short_side(im, resolution))
        imgs.append(a[..., None] if a.ndim == 2 else a)
    images = np.stack(imgs).astype(np.uint8)
    c = np.ascontiguousarray(labels[:, [3, 2, 1, 0]]).astype(np.float32)
    return Pool(images, {"c": c}, name="circuit")


def _list_image_files_recursively(data_dir):
    out = []
    for entry in sorted(os.listdir(data_dir)):
        full = os.path.join(data_dir, entry)
        if "." in entry and entry.split(".")[-1].lower() in ("jpg", "jpeg", "png", "gif"):
            out.append(full)
        elif os.path.isdir(full):
            out.extend(_list_image_files_recursively(full))
    return out


def center_crop_arr(pil_image, resolution):
    """ImageDataset.__getitem__'s resampling (image_datasets.py:147-166): BOX halving while >= 2x, BICUBIC to the short
    side, centre crop; returns u8 [res,res,3]."""
    from PIL import Image
    while min(*pil_image.size) >= 2 * resolution:
        pil_image = pil_image.resize(tuple(x // 2 for x in pil_image.size), resample=Image.BOX)
    scale = resolution / min(*pil_image.size)
    pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
    arr = np.array(pil_image.convert("RGB"))
    cy, cx = (arr.shape[0] - resolution) // 2, (arr.shape[1] - resolution) // 2
    return arr[cy:cy + resolution, cx:cx + resolution]


def read_image_folder(root, resolution, class_cond=False, shard=0, num_shards=1):
    """The "celeba" branch (image_datasets.py:92-116): x = u8/127.5 - 1; y = index of the file-name prefix before '_'."""
    from PIL import Image
    files = _list_image_files_recursively(root)
    cond = {}
    if class_cond:
        names = [os.path.basename(p).split("_")[0] for p in files]
        table = {x: i for i, x in enumerate(sorted(set(names)))}
        cond["y"] = _stride(np.asarray([table[x] for x in names], dtype=np.int64), shard, num_shards)
    files = _stride(files, shard, num_shards)
    imgs = []
    for p in files:
        with Image.open(p) as im:
            im.load()
            imgs.append(center_crop_arr(im, resolution))
    images = np.stack(imgs).astype(np.uint8) if imgs else np.zeros((0, resolution, resolution, 3), np.uint8)
    return Pool(images, cond, div=127.5, shift=-1.0, name="images")


# ---------------------------------------------------------------------------------------------------- batch feed
class Feed:
    """Infinite iterator of (x [B,C,H,W] fp32, cond dict) over a Pool with the reference DataLoader's epoch rule
    (shuffle per epoch, drop_last).  device=cuda: pool + labels resident in HBM, one gather launch per batch."""

    def __init__(self, pool, batch_size, shuffle=True, device=None, seed=0):
        if len(pool) < batch_size:
            raise ValueError(f"{pool.name}: shard has {len(pool)} items, fewer than one batch of {batch_size} (drop_last)")
        self.pool, self.batch_size, self.shuffle = pool, int(batch_size), shuffle
        self.device = th.device(device) if device is not None else th.device("cpu")
        self.gen = th.Generator().manual_seed(seed)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            from . import _lib                               # fails loudly when libcdae.so is missing
            self._lib = _lib
            n, h, w, c = pool.images.shape
            if (h * w * c) % 4:
                raise ValueError("device feed needs H*W*C to be a multiple of 4 bytes")
            self.d_images = th.from_numpy(pool.images).to(self.device)
            self.d_cond = {k: th.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in pool.cond.items()}
        self._order, self._pos = None, 0

    def _next_index(self):
        n, b = len(self.pool), self.batch_size
        if self._order is None or self._pos + b > n:         # new epoch (the tail shorter than a batch is dropped)
            self._order = th.randperm(n, generator=self.gen) if self.shuffle else th.arange(n)
            self._pos = 0
        idx = self._order[self._pos:self._pos + b]
        self._pos += b
        return idx

    def __iter__(self):
        return self

    def __next__(self):
        idx = self._next_index()
        p = self.pool
        n, h, w, c = p.images.shape
        if self.on_gpu:
            lib = self._lib
            d_idx = idx.to(self.device, non_blocking=True)
            x = th.empty((self.batch_size, c, h, w), device=self.device, dtype=th.float32, memory_format=th.channels_last)
            if c == 1:                                       # channels_last of a 1-channel tensor is ambiguous: same bytes
                x = th.empty((self.batch_size, c, h, w), device=self.device, dtype=th.float32)
            lib.check(lib.lib.cdae_gather_u8(lib.ptr(self.d_images), lib.ptr(d_idx), lib.ptr(x), self.batch_size, h * w * c,
                                             p.div, p.shift, lib.stream()))
            return x, {k: v.index_select(0, d_idx) for k, v in self.d_cond.items()}
        i = idx.numpy()
        x = p.images[i].astype(np.float32) / np.float32(p.div) + np.float32(p.shift)
        x = th.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
        return x, {k: th.from_numpy(np.ascontiguousarray(v[i])) for k, v in p.cond.items()}


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _synthetic(batch_size, image_size, class_cond, in_channels, n_vars, seed):
    rng = np.random.RandomState(seed)
    while True:
        x = th.from_numpy(rng.rand(batch_size, in_channels, image_size, image_size).astype(np.float32))
        cond = {"c": th.from_numpy(rng.rand(batch_size, n_vars).astype(np.float32))}
        if class_cond:
            cond["y"] = th.from_numpy(rng.randint(0, 10, size=(batch_size,)).astype(np.int64))
        yield x, cond


def load_data(*, data_dir, batch_size, image_size, class_cond=False, split="train", deterministic=False, device=None,
              in_channels=3, n_vars=4, seed=0):
    """Infinite generator of (batch, cond) like the reference's load_data (image_datasets.py:69-126): the dataset is
    chosen by a substring of `data_dir` ("celeba" | "morphomnist" | "pendulum" | "circuit"), each rank reads the
    [rank::world] stride.  data_dir "" / "synthetic" -> seeded synthetic tensors of the requested shape (benchmarks).
    `device`: None -> host tensors (reference behaviour); a cuda device -> HBM-resident pool + gather kernel."""
    if data_dir in ("", "synthetic", None):
        if device is not None and th.device(device).type == "cuda":
            # the same HBM-resident feed real datasets use (u8 pool + one gather launch per batch), over seeded synthetic images
            rng = np.random.RandomState(seed)
            n = max(4 * batch_size, 1024)
            cond = {"c": rng.rand(n, n_vars).astype(np.float32)}
            if class_cond:
                cond["y"] = rng.randint(0, 10, size=(n,)).astype(np.int64)
            pool = Pool(rng.randint(0, 256, size=(n, image_size, image_size, in_channels), dtype=np.uint8), cond, div=127.5, shift=-1.0,
                        name="synthetic")
            yield from Feed(pool, batch_size, shuffle=True, device=device, seed=seed)
            return
        yield from _synthetic(batch_size, image_size, class_cond, in_channels, n_vars, seed)
        return
    rank, world = _rank_world()
    if "celeba" in data_dir:
        pool, shuffle = read_image_folder(data_dir, image_size, class_cond, rank, world), not deterministic
    elif "morphomnist" in data_dir:
        pool, shuffle = read_morphomnist(data_dir, split, rank, world), True
    elif "pendulum" in data_dir:
        pool, shuffle = read_pendulum(data_dir, split, rank, world), True
    elif "circuit" in data_dir:
        pool, shuffle = read_circuit(data_dir, split, rank, world), False      # the reference does not shuffle this one
    else:
        raise ValueError(f"cannot tell the dataset from data_dir={data_dir!r} (expected celeba/morphomnist/pendulum/circuit in the path)")
    yield from Feed(pool, batch_size, shuffle=shuffle, device=device, seed=seed + rank)
