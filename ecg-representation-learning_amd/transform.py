"""
Input transforms of the reference data pipeline, FUSED into the patch-embed load.

Reference: `ecg_transformer/preprocess/transform.py` -- `Normalize` (:18-35), `TimeEndPad` (:140-154), `TimeOut` (:175-185) --
composed per record on the host in `get_ptbxl_dataset` (`preprocess/ptb_dataset.py:132-149`: Normalize, TimeEndPad(patch_size),
and TimeOut for the training split only).  Here the raw `(B, 12, L_raw)` batch goes to the device as is and the three
transforms are applied inside `ecgvit_patch_gather_transform` while the samples stream through LDS: no extra read+write of
the input tensor, no per-record numpy work on the host.  Only the TimeOut span is drawn on the host (two int32 per record), with
the same torch RNG calls the reference makes, so a seeded run masks the same spans.
"""
import torch


class FusedInputTransform:
    def __init__(self, mean, std, patch_size, timeout=False, timeout_scale=(0.0, 0.5)):
        mean = torch.as_tensor(mean, dtype=torch.float32)
        std = torch.as_tensor(std, dtype=torch.float32)
        assert mean.numel() == 12 and std.numel() == 12   # transform.py:26
        self.mean, self.inv_std = mean.contiguous(), (1.0 / std).contiguous()
        self.k = int(patch_size)
        self.timeout = bool(timeout)
        self.sampler = torch.distributions.Uniform(low=timeout_scale[0], high=timeout_scale[1])   # transform.py:178
        self._dev = None

    def padded_length(self, l_raw: int) -> int:
        """TimeEndPad: n_pad = k - (l % k)  -- a FULL extra patch when l is already a multiple of k (transform.py:150)"""
        return l_raw + (self.k - (l_raw % self.k))

    def draw_timeout(self, batch: int, l_pad: int, device):
        """per record: r ~ U(lo, hi); l_crop = round(r * L); start = randint(L - l_crop)  (transform.py:180-183), as int32 tensors"""
        starts, lens = [], []
        for _ in range(batch):
            r = self.sampler.sample().item()
            l_crop = round(r * l_pad)
            start = torch.randint(high=l_pad - l_crop, size=(1,)).item()
            starts.append(start)
            lens.append(l_crop)
        return (torch.tensor(starts, dtype=torch.int32, device=device), torch.tensor(lens, dtype=torch.int32, device=device))

    def device_stats(self, device):
        if self._dev is None or self._dev[0].device != device:
            self._dev = (self.mean.to(device), self.inv_std.to(device))
        return self._dev
