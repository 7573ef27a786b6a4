"""``load_model`` of the reference's src/script/train_ppg2mel.py:113-119 (the only part of that
script on the synthesis path, imported by generate_synthesis.py:20).  Training the PPG->mel
model is out of scope."""
from common.model import Tacotron2


def load_model(hparams):
    if hparams.fp16_run:
        raise NotImplementedError("fp16_run is not built (README.md:53 of the reference: FP16 does not work)")
    return Tacotron2(hparams).cuda()
