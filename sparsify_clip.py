#!/usr/bin/env python
"""Experiment runner with the reference's command line (reference sparsify_clip.py:1127-1156):

    python sparsify_clip.py --config <yaml file | directory of yaml files> --device <gpu id>

Every experiments_configs/*.yaml and ablatation_configs/*.yaml of the reference is read and run unchanged (they all say model
"RN50": the ModifiedResNet-50 tower of sparsify_clip_amd/resnet.py).  The optional overrides select the ViT configurations and
batch sizes of BASELINE.json without editing the files.  Under torchrun (WORLD_SIZE > 1) the run is data-parallel over the node's GPUs and --device is
replaced by LOCAL_RANK.  The loss/schedule functions of the reference are importable from this module by name.
"""
import argparse
import json
import sys

from sparsify_clip_amd import dist as D
from sparsify_clip_amd.config import config_files, load_config
from sparsify_clip_amd.losses import (centroid_alignment_loss, compute_centroids, compute_centroids_only, contrastive_loss,  # noqa: F401
                                      contrastive_loss_roberta, lalign_loss, lunif_loss, random_alignment_loss, sparsify_loss)
from sparsify_clip_amd.schedules import get_alpha, get_beta, get_cosine_schedule_with_warmup  # noqa: F401
from sparsify_clip_amd.train import evaluate_model, main, set_seed, train_model  # noqa: F401
from sparsify_clip_amd.uniformity import (compute_gap, compute_mean_angular_value_of_a_modality, mean_distance_of_true_pairs,  # noqa: F401
                                          uniformity)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Run the experiment with a config.yaml file")
    p.add_argument("--config", type=str, required=True, help="Path to the yaml config file or to a folder containing multiple config files")
    p.add_argument("--device", type=int, required=True, help="GPU id to use")
    # optional overrides on top of the unchanged YAML (default: use the YAML's value)
    p.add_argument("--model", type=str, default=None, help="RN50 (what the YAMLs say), ViT-B-32, ViT-L-14")
    p.add_argument("--batch-size", type=int, default=None, help="global batch size")
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--num-train-samples", type=int, default=None)
    p.add_argument("--num-test-samples", type=int, default=None)
    p.add_argument("--eval-batch-size", type=int, default=None)
    p.add_argument("--steps-per-epoch", type=int, default=None)
    p.add_argument("--precision", choices=["bf16", "fp32"], default=None)
    p.add_argument("--micro-batch", type=int, default=None, help="run every step through Trainer.step_cached with this micro-batch: the same step "
                   "(loss over the whole batch) for batches whose activations do not fit the GPU; ViT models, one process")
    return p.parse_args(argv)


def run(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = D.init_process_group()
    device_id = local_rank if world > 1 else args.device
    overrides = {"model": args.model, "batch_size": args.batch_size, "epochs": args.epochs, "num_train_samples": args.num_train_samples,
                 "num_test_samples": args.num_test_samples, "eval_batch_size": args.eval_batch_size, "steps_per_epoch": args.steps_per_epoch,
                 "precision": args.precision, "micro_batch": args.micro_batch}
    results = {}
    for path in config_files(args.config):
        config = load_config(path, device_id, overrides)
        if config is None:      # empty YAML (all_experiments.yaml): skipped, the reference crashes here (:1152)
            continue
        results[config["run_name"]] = main(config)
        if rank == 0:
            print(json.dumps({"run": config["run_name"], **results[config["run_name"]]}))
    return results


if __name__ == "__main__":
    run(sys.argv[1:])
