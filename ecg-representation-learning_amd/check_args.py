"""
Enum-style argument validation for the hot path's entry points.

Mirrors the accepted value sets of the reference's `CheckArg` (`ecg_transformer/util/check_args.py:18-20`) for the
arguments that reach the model / train step: `model_name`, `optimizer`, `schedule`.  Raises `ValueError` on a
mismatch, which is what the reference's `check_mismatch` (:25-28) is written to do.  (Quirk, recorded in
tests/golden/host_contract.json: as imported, the reference's circular import leaves `logi` undefined inside
`check_args`, so its raise site actually dies with NameError; the intended ValueError is what we implement.)
"""

MODEL_NAMES = ['ecg-vit-debug', 'ecg-vit-tiny', 'ecg-vit-small', 'ecg-vit-base', 'ecg-vit-large']
OPTIMIZERS = ['Adam', 'AdamW']
SCHEDULES = ['constant', 'cosine']


class CheckArg:
    model_names, optimizer, schedule = MODEL_NAMES, OPTIMIZERS, SCHEDULES

    @staticmethod
    def check_mismatch(arg_type, arg_value, expected_values):
        if arg_value not in expected_values:
            raise ValueError(f'Unexpected {arg_type}: expect one of {expected_values}, got {arg_value}')

    def __init__(self):
        self._checks = dict(
            model_name=lambda v: CheckArg.check_mismatch('Model Name', v, MODEL_NAMES),
            optimizer=lambda v: CheckArg.check_mismatch('Optimizer', v, OPTIMIZERS),
            schedule=lambda v: CheckArg.check_mismatch('Schedule', v, SCHEDULES),
        )

    def __call__(self, **kwargs):
        for k, v in kwargs.items():
            self._checks[k](v)


ca = CheckArg()
