"""
Record feeding for the train / eval loops: N x 12 x L record arrays -> device batches.

Mirror of what the reference does around the model: `EcgDataset` / `PtbxlDataset` (`ecg_transformer/preprocess/dataset.py:22-99`,
`preprocess/ptb_dataset.py:53-78`: an HDF5 'data' array of float64 records indexed by PTB-XL csv row, labels as lists of code
ids -> 71-wide multi-hot), `get_ptbxl_splits` (`ptb_dataset.py:101-133`: `strat_fold` < 9 train, 9 eval, 10 test) and the
`DataLoader(batch_size, shuffle, pin_memory=True, num_workers=0)` of `PtbxlDataModule` (`:81-98`).  The reference transforms every
record on the host, one numpy call at a time, in the training process (HDF5 handles cannot be pickled: `num_workers=0`), which caps
it far below the 5-6 k records/s the MI355X step consumes.  Here the host only GATHERS raw rows (one fancy-index per batch, in a
background thread) into pinned memory and starts an asynchronous H2D copy on a side stream; Normalize / TimeEndPad / TimeOut run on
the device inside the patch-embed load (`transform.FusedInputTransform`), labels are expanded to multi-hot once, up front.

HDF5 needs `h5py` (absent from this image: the import is attempted only when an `.h5/.hdf5` path is given); `.npy` files are
memory-mapped; arrays are used as they are.
"""
import os
import threading
from collections import namedtuple

import numpy as np
import torch

PtbxlSplits = namedtuple('PtbxlSplits', ['train', 'eval', 'test'])   # index arrays into the record store (0-indexed csv rows)
N_CLASS = 71


def open_records(source):
    """(N, C, L) record array: ndarray / memmap as is, '*.npy' memory-mapped, '*.h5|*.hdf5' -> its 'data' dataset (needs h5py)"""
    if isinstance(source, str):
        ext = os.path.splitext(source)[1].lower()
        if ext == '.npy':
            return np.load(source, mmap_mode='r')
        if ext in ('.h5', '.hdf5'):
            try:
                import h5py
            except ImportError as e:
                raise ImportError('reading the reference\'s HDF5 record files needs h5py, which this environment does not have; '
                                  'export the "data" dataset to .npy once and pass that path') from e
            return h5py.File(source, 'r')['data']
        raise ValueError(f'unsupported record file {source!r}')
    if getattr(source, 'ndim', 0) != 3:
        raise ValueError('records must be an (N, C, L) array')
    return source


def ptbxl_splits(strat_fold, n_sample=None) -> PtbxlSplits:
    """`get_ptbxl_splits`: folds 1-8 train, 9 eval, 10 test; `n_sample` keeps the first rows of each split (ptb_dataset.py:111-113)"""
    f = np.asarray(strat_fold)
    tr, vl, ts = np.nonzero(f < 9)[0], np.nonzero(f == 9)[0], np.nonzero(f == 10)[0]
    if n_sample is not None:
        tr, vl, ts = tr[:n_sample], vl[:n_sample], ts[:n_sample]
    else:
        assert len(tr) + len(vl) + len(ts) == len(f)   # the reference's own sanity check: folds are in [1, 10]
    return PtbxlSplits(tr, vl, ts)


def lbs2multi_hot(labels, n_class=N_CLASS):
    """`PtbxlDataset.lbs2multi_hot(return_float=True)` for a whole split at once: list of code-id lists -> (n, n_class) float32"""
    out = np.zeros((len(labels), n_class), dtype=np.float32)
    for i, lb in enumerate(labels):
        out[i, list(lb)] = 1.0
    return out


class DeviceFeeder:
    """Iterate `dict(sample_values=(b, C, L) f32, labels=(b, K) f32)` device batches over `records[idxs]` (raw, untransformed).

    Double-buffered: while the step consumes batch i, a worker thread gathers batch i+1 from the (memory-mapped) store into pinned
    memory (float64 -> float32 as the reference's `__getitem__` does) and queues its H2D copy on a side stream; the consumer's stream
    waits on the copy's event only.  `shuffle` permutes with `torch.randperm` per epoch (seeded); `rank/world` take a contiguous,
    equally long shard of every epoch's order (wrapped around when n is not a multiple of the world size: `pad=True`, what the
    collective train step needs).  `pad=False` (EVALUATION) deals the exact records instead -- the last ranks' shards are up to
    world-1 records shorter and nothing is counted twice; `HipEvaluator(gather=True)` all-gathers unequal shards by their true sizes.
    On a CPU-only host (tests) it degrades to synchronous host tensors.
    """

    def __init__(self, records, idxs, labels_multi_hot, batch_size, shuffle=False, seed=77, device=None, rank=0, world=1,
                 drop_last=False, pad=True):
        self.rec = open_records(records)
        self.idxs = np.asarray(idxs, dtype=np.int64)
        self.labels = np.ascontiguousarray(labels_multi_hot, dtype=np.float32)
        assert len(self.idxs) == len(self.labels)
        self.bsz, self.shuffle, self.seed, self.drop_last = int(batch_size), shuffle, seed, drop_last
        self.rank, self.world, self.pad = rank, world, bool(pad)
        self.device = torch.device(device) if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.on_gpu = self.device.type == 'cuda'
        self.epoch = 0
        _, self.C, self.L = self.rec.shape

    def __len__(self):
        n = self._shard_len()
        if not self.pad:   # exact shard of this rank (evaluation): may be shorter than the others', or empty
            n = max(0, min(n, len(self.idxs) - self.rank * n))
        return n // self.bsz if self.drop_last else (n + self.bsz - 1) // self.bsz

    def _shard_len(self):
        """records per rank and epoch: EQUAL on every rank (ceil(n / world), the epoch's order wraps around to fill the last shard, as
        torch's DistributedSampler pads), so that every rank yields the same number of batches of the same sizes -- the train step's
        gradient all-reduce is collective, a rank with one batch fewer would hang the others"""
        n = len(self.idxs)
        return (n + self.world - 1) // self.world

    def _order(self):
        n = len(self.idxs)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g).numpy()
        else:
            perm = np.arange(n)
        per = self._shard_len()
        if self.pad and per * self.world > n:
            perm = np.concatenate([perm, perm[:per * self.world - n]])
        return perm[self.rank * per:(self.rank + 1) * per]

    def _gather(self, rows, x_buf, y_buf):
        b = len(rows)
        src = self.idxs[rows]
        # HDF5 fancy indexing wants STRICTLY increasing indices (and a memmap reads best that way): read each distinct record once, in
        # increasing order, and scatter -- with pad=True the epoch's order wraps around, so a batch may name a record twice
        uniq, inv = np.unique(src, return_inverse=True)
        tmp = np.asarray(self.rec[uniq])
        x_buf[:b].numpy()[...] = tmp[inv]                        # float64 -> float32 happens in this assignment
        y_buf[:b].numpy()[...] = self.labels[rows]
        return b

    def __iter__(self):
        order = self._order()
        self.epoch += 1
        nb = len(self)
        bounds = [(i * self.bsz, min((i + 1) * self.bsz, len(order))) for i in range(nb)]
        if not self.on_gpu:
            for lo, hi in bounds:
                x = torch.empty(hi - lo, self.C, self.L)
                y = torch.empty(hi - lo, self.labels.shape[1])
                self._gather(order[lo:hi], x, y)
                yield dict(sample_values=x, labels=y)
            return
        pin = [(torch.empty(self.bsz, self.C, self.L).pin_memory(), torch.empty(self.bsz, self.labels.shape[1]).pin_memory()) for _ in range(2)]
        dev = [(torch.empty(self.bsz, self.C, self.L, device=self.device), torch.empty(self.bsz, self.labels.shape[1], device=self.device))
               for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=self.device)
        ready = [torch.cuda.Event(), torch.cuda.Event()]          # H2D copy of slot s has completed
        consumed = [torch.cuda.Event(), torch.cuda.Event()]       # the consumer's kernels reading slot s have been queued before this event
        sizes = [0, 0]

        def produce(i):
            s = i & 1
            lo, hi = bounds[i]
            consumed[s].synchronize()                             # the device no longer reads dev[s] (2 batches ago)
            sizes[s] = self._gather(order[lo:hi], *pin[s])
            with torch.cuda.stream(copy_stream):
                dev[s][0][:sizes[s]].copy_(pin[s][0][:sizes[s]], non_blocking=True)
                dev[s][1][:sizes[s]].copy_(pin[s][1][:sizes[s]], non_blocking=True)
                ready[s].record(copy_stream)

        for e in consumed:
            e.record()
        worker = None
        if nb:
            produce(0)
        for i in range(nb):
            if worker is not None:
                worker.join()
            if i + 1 < nb:
                worker = threading.Thread(target=produce, args=(i + 1,), daemon=True)
                worker.start()
            else:
                worker = None
            s = i & 1
            torch.cuda.current_stream().wait_event(ready[s])
            yield dict(sample_values=dev[s][0][:sizes[s]], labels=dev[s][1][:sizes[s]])
            consumed[s].record()                                  # everything the caller queued on its stream for this batch precedes this
        if worker is not None:
            worker.join()
