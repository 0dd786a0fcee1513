"""
ecg_representation_learning_amd -- MI355X-native (gfx950) drop-in for the ECG-ViT train step of
StefanHeng/ECG-Representation-Learning (`ecg_transformer.models`): same `EcgVitConfig` / `EcgVit.forward ->
ModelOutput(loss, logits)` / train-step surface, executed by hand-written HIP kernels behind a C-ABI
(`include/ecgvit_hip.h`, `csrc/`).  Importing the package never needs a GPU; running the model does, and
fails loudly when the HIP library is absent (no CPU / eager fallback by design).
"""
from .check_args import ca, CheckArg
from .ecg_vit import EcgVitConfig, EcgVit, ModelOutput, HipViT, MaskedEcgVit, load_trained
from .train import get_train_args, lr_multiplier, HipTrainStep, clip_grad_norm_
from .transform import FusedInputTransform
from .metrics import get_accuracy, eval_counts, HipEvaluator
from .feed import DeviceFeeder, ptbxl_splits, lbs2multi_hot, open_records
from . import hip
from . import ddp
from . import workload

__all__ = ['ca', 'CheckArg', 'EcgVitConfig', 'EcgVit', 'ModelOutput', 'HipViT', 'MaskedEcgVit', 'load_trained', 'get_train_args', 'lr_multiplier',
           'HipTrainStep', 'clip_grad_norm_', 'FusedInputTransform', 'get_accuracy', 'eval_counts', 'HipEvaluator', 'DeviceFeeder', 'ptbxl_splits', 'lbs2multi_hot', 'open_records', 'hip', 'ddp', 'workload']
