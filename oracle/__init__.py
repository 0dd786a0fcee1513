"""
oracle/ -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (PyTorch-eager fp32 / numpy) restatement of the reference hot path:
`EcgVit.forward` -> loss, and the `MyTrainer.train` step body, of
StefanHeng/ECG-Representation-Learning (reference files cited per function).

Who may import this package: `tests/`, `__graft_entry__.smoke()`, and the
`cpu_baseline` leg of `bench.py` -- and there only as the checker / the timed
CPU baseline.  `ecg_representation_learning_amd` (the product) never does; it
raises if its HIP library is missing instead of falling back to this code.

PARITY STATUS -- read before trusting any number:
  * The reference's OWN code on this path (`EcgVitConfig`, `EcgVit.__init__/forward`,
    BCE loss, `get_train_args`, HF warm-up schedules) is pinned: `oracle/make_golden.py`
    executes the real reference files (UI/IO-only imports stubbed) and the committed
    fixtures under `tests/golden/` hold their outputs.
  * The transformer arithmetic itself lives in third-party `vit-pytorch==0.33.2`
    (reference `requirements.txt:174`), which is NOT vendored under /root/reference,
    not installed and not fetchable.  `oracle/vit_oracle.py` restates its published
    algorithm (pre-LN blocks, bias-free fused QKV, scale after QK^T, exact-erf GELU,
    LayerNorm eps 1e-5, CLS pooling, LN+Linear head).  The reference holds no test,
    golden vector or known-answer value for it, so for that part:
        **parity unpinned** (structural pins only: state_dict key layout,
        base param count 85 705 799, logits shape, meta strings).
  * The masked pre-train objective (SimMIM-style) does not exist in the reference at
    all; its oracle restates the build's own definition: **parity unpinned**.
"""
