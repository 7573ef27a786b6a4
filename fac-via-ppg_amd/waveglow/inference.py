"""Stand-alone mel -> wav CLI -- drop-in for src/waveglow/inference.py: a list of mel ``.pt`` files
([n_mel, T] tensors, e.g. written by ``python -m waveglow.mel2samp``), one ``*_synthesis.wav`` (int16)
per file.  Files are synthesised as ONE padded batch with per-file lengths (same audio as one call per
file; the reference loops one by one, inference.py:43-56).  ``--is_fp16`` is refused: the HIP path is fp32."""
import argparse
import os

import torch
from scipy.io.wavfile import write

from waveglow.mel2samp import MAX_WAV_VALUE, files_to_list


def main(mel_files, waveglow_path, sigma, output_dir, sampling_rate, is_fp16, batch_size=16):
    mel_files = files_to_list(mel_files)
    waveglow = torch.load(waveglow_path, weights_only=False)['model']
    waveglow = waveglow.remove_weightnorm(waveglow)
    waveglow.cuda().eval()
    if is_fp16:
        # The reference casts the module and every mel to half here (inference.py:38-48, through apex) and runs cuDNN's half
        # kernels.  This library computes on the fp32 MFMA path only, so the half branch is run as "the reference's VALUES, this
        # library's arithmetic": parameters, buffers and mels are rounded to fp16 and every product is then formed and
        # accumulated in fp32 -- at least the precision of the reference's run, and within fp16 round-off of it.
        with torch.no_grad():
            for t in list(waveglow.parameters()) + list(waveglow.buffers()):
                if t.is_floating_point():
                    t.copy_(t.half().float())
    hop = waveglow.upsample.stride[0]
    os.makedirs(output_dir, exist_ok=True)
    for i0 in range(0, len(mel_files), batch_size):
        paths = mel_files[i0:i0 + batch_size]
        mels = [torch.load(p, weights_only=False).float() for p in paths]
        if is_fp16:
            mels = [m.half().float() for m in mels]
        lens = [m.shape[1] for m in mels]
        batch = torch.zeros(len(mels), mels[0].shape[0], max(lens))
        for b, m in enumerate(mels):
            batch[b, :, :lens[b]] = m
        with torch.no_grad():
            audio = MAX_WAV_VALUE * waveglow.infer(batch.cuda(), sigma=sigma, lengths=lens if len(lens) > 1 else None)
        audio = audio.cpu().numpy()
        for b, p in enumerate(paths):
            name = os.path.splitext(os.path.basename(p))[0]
            audio_path = os.path.join(output_dir, "{}_synthesis.wav".format(name))
            write(audio_path, sampling_rate, audio[b, :lens[b] * hop].astype('int16'))
            print(audio_path)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('-f', "--filelist_path", required=True)
    parser.add_argument('-w', '--waveglow_path', help='Path to waveglow decoder checkpoint with model')
    parser.add_argument('-o', "--output_dir", required=True)
    parser.add_argument("-s", "--sigma", default=1.0, type=float)
    parser.add_argument("--sampling_rate", default=22050, type=int)
    parser.add_argument("--is_fp16", action="store_true")
    args = parser.parse_args()
    main(args.filelist_path, args.waveglow_path, args.sigma, args.output_dir, args.sampling_rate, args.is_fp16)
