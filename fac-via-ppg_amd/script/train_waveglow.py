"""WaveGlow training loop -- drop-in for src/script/train_waveglow.py.

Same config file (waveglow/config.json: train/data/dist/waveglow sections), checkpoint format
(``{'model': <pickled WaveGlow>, 'iteration', 'optimizer', 'learning_rate'}``), Adam optimiser and
data-parallel scheme (one process per GPU, gradients averaged by waveglow.distributed).  The step
itself -- WaveGlow.forward, WaveGlowLoss, backward -- runs on libfacppg_hip through the model's
autograd nodes; the mel features of each batch are computed on the GPU from the audio segments.

  PYTHONPATH=fac-via-ppg_amd python -m script.train_waveglow -c fac-via-ppg_amd/waveglow/config.json
"""
import argparse
import json
import os

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from waveglow.distributed import (GradientExchange, init_distributed, apply_gradient_allreduce, broadcast_parameters,
                                  reduce_tensor)
from waveglow.glow import WaveGlow, WaveGlowLoss
from waveglow.graphed import GraphedTrainStep
from waveglow.mel2samp import Mel2Samp
from waveglow.optim import Adam


def load_checkpoint(checkpoint_path, model, optimizer):
    """train_waveglow.py:45-54"""
    assert os.path.isfile(checkpoint_path)
    checkpoint_dict = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    iteration = checkpoint_dict['iteration']
    optimizer.load_state_dict(checkpoint_dict['optimizer'])
    model.load_state_dict(checkpoint_dict['model'].state_dict())
    print("Loaded checkpoint '{}' (iteration {})".format(checkpoint_path, iteration))
    return model, optimizer, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, filepath, waveglow_config):
    """train_waveglow.py:56-64: the whole module is pickled (load_waveglow_model relies on it)."""
    print("Saving model and optimizer state at iteration {} to {}".format(iteration, filepath))
    model_for_saving = WaveGlow(**waveglow_config)
    model_for_saving.load_state_dict(model.state_dict())
    torch.save({'model': model_for_saving, 'iteration': iteration, 'optimizer': optimizer.state_dict(),
                'learning_rate': learning_rate}, filepath)


def train_step(model, criterion, optimizer, mel, audio, num_gpus=1):
    """One optimisation step (train_waveglow.py:121-134); returns the (rank-averaged) loss."""
    model.zero_grad()
    loss = criterion(model((mel, audio)))
    reduced_loss = reduce_tensor(loss.data, num_gpus).item() if num_gpus > 1 else loss.item()
    loss.backward()          # waveglow.distributed's hook averages the gradients over ranks
    optimizer.step()
    return reduced_loss


def train(num_gpus, rank, group_name, output_directory, epochs, learning_rate, sigma, iters_per_checkpoint, batch_size, seed,
          checkpoint_path, data_config, dist_config, waveglow_config, max_iterations=None, train_precision=None, hip_graph=None,
          grad_buckets=3, grad_dtype="fp32"):
    """train_waveglow.py:66-147.  ``train_precision`` (optional key of the config's train section): 'fp32' (default, the
    reference's arithmetic) or 'bf16' (bf16 MFMA operands, fp32 accumulation / master weights / gradients); the
    FACPPG_TRAIN_PRECISION environment variable sets the default.  ``hip_graph`` (optional key): replay the step as one
    captured HIP graph (waveglow.graphed); default: on for bf16 (FACPPG_TRAIN_GRAPH=0 turns it off), off for fp32.
    ``grad_buckets`` / ``grad_dtype`` (optional keys, data parallel only): number of flat gradient buckets reduced
    asynchronously behind the backward pass, and what the links carry ('fp32', or 'bf16' = half the bytes)."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    if num_gpus > 1:
        init_distributed(rank, num_gpus, group_name, **dist_config)
    criterion = WaveGlowLoss(sigma)
    model = WaveGlow(**waveglow_config).cuda()
    if train_precision is not None:
        model.train_precision = train_precision
    if hip_graph is None:
        hip_graph = model.train_precision == "bf16" and os.environ.get("FACPPG_TRAIN_GRAPH", "1") != "0"
    exchange = None
    comm_dtype = {"fp32": None, "bf16": torch.bfloat16}[grad_dtype]
    if num_gpus > 1:
        # eager: buckets are reduced from gradient hooks while the backward pass runs; graphed: the stepper installs the same
        # hooks and captures the collectives INTO the graph (waveglow.graphed; if this stack cannot capture them, the bucket
        # pipeline follows each replay instead)
        if hip_graph:
            broadcast_parameters(model, 0)
            exchange = GradientExchange(model, n_buckets=grad_buckets, grad_dtype=comm_dtype)
        else:
            model = apply_gradient_allreduce(model, n_buckets=grad_buckets, grad_dtype=comm_dtype)
    # torch.optim.Adam's state and arithmetic, ONE HIP launch over all 938 parameters (waveglow.optim); inside the graph
    # when there is no gradient exchange
    optimizer = Adam(model.parameters(), lr=learning_rate)
    stepper = None
    if hip_graph:
        seg = data_config["segment_length"]
        stepper = GraphedTrainStep(model, criterion, optimizer, exchange=exchange,
                                   expected_shapes=((batch_size, waveglow_config["n_mel_channels"], seg // data_config["hop_length"] + 1),
                                                    (batch_size, seg)))
    iteration = 0
    if checkpoint_path != "":
        model, optimizer, iteration = load_checkpoint(checkpoint_path, model, optimizer)
        iteration += 1
    trainset = Mel2Samp(audio_only=True, **data_config)
    train_sampler = DistributedSampler(trainset) if num_gpus > 1 else None
    train_loader = DataLoader(trainset, num_workers=0, shuffle=train_sampler is None, sampler=train_sampler,
                              batch_size=batch_size, pin_memory=True, drop_last=True)
    if rank == 0:
        os.makedirs(output_directory, exist_ok=True)
        print("output directory", output_directory)
    model.train()
    epoch_offset = max(0, int(iteration / max(1, len(train_loader))))
    for epoch in range(epoch_offset, epochs):
        print("Epoch: {}".format(epoch))
        for audio in train_loader:
            audio = audio.cuda(non_blocking=True)
            with torch.no_grad():
                mel = trainset.mel_batch(audio)           # GPU STFT -> mel, no host round trip
            if stepper is not None:
                loss = stepper(mel, audio)
                reduced_loss = reduce_tensor(loss, num_gpus).item() if num_gpus > 1 else loss.item()
            else:
                reduced_loss = train_step(model, criterion, optimizer, mel, audio, num_gpus)
            print("{}:\t{:.9f}".format(iteration, reduced_loss))
            if iteration % iters_per_checkpoint == 0 and rank == 0:
                save_checkpoint(model, optimizer, learning_rate, iteration, "{}/waveglow_{}".format(output_directory, iteration),
                                waveglow_config)
            iteration += 1
            if max_iterations is not None and iteration >= max_iterations:
                return model


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=str, default=os.path.join(os.path.dirname(__file__), '..', 'waveglow', 'config.json'))
    parser.add_argument('-r', '--rank', type=int, default=None)
    parser.add_argument('-g', '--group_name', type=str, default=None)
    parser.add_argument('--num_gpus', type=int, default=None,
                        help='world size (waveglow.distributed.main passes it, distributed.py:151-152); default: every visible GPU')
    args = parser.parse_args(argv)
    with open(args.config) as f:
        config = json.load(f)
    train_config, data_config = config["train_config"], config["data_config"]
    dist_config, waveglow_config = dict(config["dist_config"]), config["waveglow_config"]
    rank = dist_config.pop("rank") if args.rank is None else (dist_config.pop("rank"), args.rank)[1]
    group_name = dist_config.pop("group_name") if args.group_name is None else (dist_config.pop("group_name"), args.group_name)[1]
    num_gpus = torch.cuda.device_count() if args.num_gpus is None else args.num_gpus
    if num_gpus > 1 and group_name == '':
        print("WARNING: Multiple GPUs detected but no distributed group set")
        print("Only running 1 GPU.  Use distributed launch (one process per GPU) for multiple GPUs")
        num_gpus = 1
    if num_gpus == 1 and rank != 0:
        raise Exception("Doing single GPU training on rank > 0")
    os.makedirs(train_config["output_directory"] or ".", exist_ok=True)
    with open(os.path.join(train_config["output_directory"] or ".", 'config.json'), 'w') as writer:
        json.dump(config, writer)
    train(num_gpus, rank, group_name, data_config=data_config, dist_config=dist_config, waveglow_config=waveglow_config,
          **train_config)


if __name__ == "__main__":
    main()
