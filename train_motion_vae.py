"""Training entrypoint — drop-in for the reference's train_motion_vae.py (same flags, same loop shape:
train_motion_vae.py:20-163 there), running the MI355X-native engine.

    python train_motion_vae.py --config configs/len64_no_aug_hm_vae.yaml --output_path ./out [--batch_size 256]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_motion_vae.py --config ... --multigpus

Extra flags (not in the reference): --batch_size / --max_iter override the YAML, --data_folder points at a directory of
`T x 579` npy sequences (the reference's on-disk format); without it AMASS-shaped synthetic batches are used (the AMASS
data itself is not redistributable).  `--multigpus` means one process per GPU with an RCCL all-reduce of the flat
gradient bucket (the reference's flag wrapped the model in nn.DataParallel).
"""
import argparse
import os
import shutil
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hm_vae_amd import ddp  # noqa: E402
from hm_vae_amd.trainer_motion_vae import Trainer  # noqa: E402
from hm_vae_amd.utils_common import get_config, make_result_folders, make_writer, write_loss  # noqa: E402
from hm_vae_amd.utils_motion_vae import get_train_loaders_all_data_seq  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='', help='configuration file for training and testing')
    parser.add_argument('--output_path', type=str, default='./', help="outputs path")
    parser.add_argument('--test_batch_size', type=int, default=10)
    parser.add_argument('--multigpus', action="store_true")
    parser.add_argument("--resume", action="store_true")
    parser.add_argument('--test_model', type=str, default='', help="trained model for evaluation")
    parser.add_argument('--batch_size', type=int, default=0, help="override the YAML batch_size (per process)")
    parser.add_argument('--max_iter', type=int, default=0, help="override the YAML max_iter")
    parser.add_argument('--data_folder', type=str, default='', help="directory of T x 579 npy sequences")
    parser.add_argument('--graph', action="store_true", help="replay the training step from a hipGraph (not in the reference)")
    opts = parser.parse_args()

    config = get_config(opts.config)
    if opts.batch_size:
        config['batch_size'] = opts.batch_size
    if opts.max_iter:
        config['max_iter'] = opts.max_iter
    max_iter = config['max_iter']

    rank, world, local = ddp.init_from_env() if opts.multigpus else (0, 1, 0)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    config['gpus'] = world

    trainer = Trainer(config)
    trainer.cuda()
    ddp.broadcast_parameters([p for p in trainer.model.parameters() if p.requires_grad])

    train_loader, val_loader, test_loader = get_train_loaders_all_data_seq(config, device, opts.data_folder or None)

    model_name = os.path.splitext(os.path.basename(opts.config))[0]
    output_directory = os.path.join(opts.output_path + "/outputs", model_name)
    checkpoint_directory, image_directory = make_result_folders(output_directory)
    train_writer = None
    if rank == 0:
        shutil.copyfile(opts.config, os.path.join(output_directory, 'config.yaml'))
        train_writer = make_writer(os.path.join(output_directory, "logs"))

    iterations = trainer.resume(checkpoint_directory, hp=config, multigpus=opts.multigpus) if opts.resume else 0
    if opts.test_model:
        trainer.load_ckpt(opts.test_model)

    while True:
        for it, input_data in enumerate(train_loader):
            if world > 1 and input_data[0].shape[0] == config['batch_size'] * world:
                input_data = ddp.shard_batch(input_data, rank, world)
            if opts.graph:
                out = trainer.gen_update_graph(input_data, config, iterations, opts.multigpus)
            else:
                out = trainer.gen_update(input_data, config, iterations, opts.multigpus)
            if it % 50 == 0 and rank == 0:
                print('Training: Total loss: %.4f, KL loss: %.8f, Rec 6D loss: %.4f, Rec Rot loss: %.4f, Rec Pose loss: %.4f, '
                      'Rec joint pos loss: %.4f, Rec root v loss: %.4f, Rec linear v loss: %.4f, Rec angular v loss: %.4f' % out)
            if val_loader is not None and (iterations + 1) % config['validation_iter'] == 0:
                with torch.no_grad():
                    for val_it, val_data in enumerate(val_loader):
                        if val_it >= 50:
                            break
                        vout = trainer.gen_update(val_data, config, iterations, opts.multigpus, validation_flag=True)
                        if rank == 0 and val_it == 0:
                            print('Val Total loss: %.4f, Val KL loss: %.8f, Val Rec 6D loss: %.4f, Val Rec Rot loss: %.4f, '
                                  'Val Rec Pose loss: %.4f' % vout[:5])
            if test_loader is not None and (iterations + 1) % config['image_save_iter'] == 0 and rank == 0:
                with torch.no_grad():
                    for test_it, test_data in enumerate(test_loader):
                        if test_it >= opts.test_batch_size:
                            break
                        gt_seq, mean_seq, sampled_seq, _ = trainer.gen_seq(test_data, config, iterations)
                        # (T, B, 24, 3) joint positions; the reference renders matplotlib animations here
                        torch.save({'gt': gt_seq.cpu(), 'mean': mean_seq.cpu(), 'sampled': sampled_seq.cpu()},
                                   os.path.join(image_directory, 'seq_%08d_%02d.pt' % (iterations + 1, test_it)))
            if (iterations + 1) % config['log_iter'] == 0 and rank == 0:
                print("Iteration: %08d/%08d" % (iterations + 1, max_iter))
                write_loss(iterations, trainer, train_writer)
            if (iterations + 1) % config['snapshot_save_iter'] == 0:
                trainer.save(checkpoint_directory, iterations, opts.multigpus)
                if rank == 0:
                    print('Saved model at iteration %d' % (iterations + 1))
            iterations += 1
            if iterations >= max_iter:
                if rank == 0:
                    print("Finish Training")
                return


if __name__ == "__main__":
    main()
