"""Accent-conversion synthesis CLI -- the drop-in for src/script/generate_synthesis.py.

The contract kept from the reference (generate_synthesis.py:29-102) is its interface: the four required flags,
the fixed settings (16 kHz, sigma 0.6, denoiser 'zeros' at 0.005, no clipping), what ``debug.log`` says, and
``ac.wav`` (float32 [N, 1]) in the output directory -- or a logged warning and exit 0 when the teacher utterance
is missing.  The work itself is ``facppg.pipeline.Synthesizer``: models on libfacppg_hip.so, PPG read from a
precomputed .npy (common.data_utils.get_ppg).

  PYTHONPATH=fac-via-ppg_amd python -m script.generate_synthesis --ppg2mel_model tacotron.pt \\
      --waveglow_model waveglow.pt --teacher_utterance_path utt.wav --output_dir out/
"""
import argparse
import logging
import os

import ppg
from facppg.pipeline import Synthesizer

SETTINGS = {"is_clip": False, "fs": 16000, "sigma": 0.6, "denoiser_strength": 0.005, "denoiser_mode": "zeros"}

FLAGS = (("--ppg2mel_model", "Path to the PPG-to-Mel model."),
         ("--waveglow_model", "Path to the WaveGlow model."),
         ("--teacher_utterance_path", "Path to a native speaker recording (its PPG: <path>.ppg.npy, or a .npy path)."),
         ("--output_dir", "Output dir, will save the audio and log info."))


def parse(argv=None):
    parser = argparse.ArgumentParser(description="Generate accent conversion speech using pre-trained models.")
    for flag, text in FLAGS:
        parser.add_argument(flag, type=str, required=True, help=text)
    return parser.parse_args(argv)


def log_settings(args):
    """The reference's debug.log header (generate_synthesis.py:49,64-71), same wording and formats."""
    logging.info("Output dir: %s", args.output_dir)
    for fmt, value in (("Tacotron: %s", args.ppg2mel_model), ("Waveglow: %s", args.waveglow_model), ("AM: %s", "SI model"),
                       ("is_clip: %d", SETTINGS["is_clip"]), ("Fs: %d", SETTINGS["fs"]), ("Sigma: %f", SETTINGS["sigma"]),
                       ("Denoiser strength: %f", SETTINGS["denoiser_strength"]),
                       ("Denoiser mode: %s", SETTINGS["denoiser_mode"])):
        logging.debug(fmt, value)


def main(argv=None):
    args = parse(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    logging.basicConfig(filename=os.path.join(args.output_dir, "debug.log"), level=logging.DEBUG, force=True)
    log_settings(args)
    synthesizer = Synthesizer(args.ppg2mel_model, args.waveglow_model, denoiser_mode=SETTINGS["denoiser_mode"])
    utt = args.teacher_utterance_path
    if synthesizer.has_utterance(utt):
        logging.info("Perform AC on %s", utt)
        synthesizer.synthesize_file(utt, os.path.join(args.output_dir, "ac.wav"), SETTINGS["fs"], SETTINGS["sigma"],
                                    SETTINGS["denoiser_strength"], ppg_deps=ppg.DependenciesPPG())
    else:
        logging.warning("Missing %s", utt)
    logging.info("Done!")


if __name__ == "__main__":
    main()
