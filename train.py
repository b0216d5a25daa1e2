#!/usr/bin/env python
"""train.py -- entry point with the reference's CLI (reference train.py:72-193):
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py --config configs/MAGMA_v1.yml
(the reference used the `deepspeed` launcher; here one process per GPU, RCCL over xGMI).
Dataset directories in the reference's image_data/*.json layout are read by magma_amd.datasets.ImgCptDataset (a list
of directories -> ConcatDataset, eval_dataset_dir: null -> eval_dataset_pct of the train set, reference train.py:34-66);
only the literal "synthetic" selects SyntheticImgCptDataset -- a missing directory raises."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across processes on this driver); before the runtime loads
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from magma_amd import Magma  # noqa: E402
from magma_amd.datasets import SyntheticImgCptDataset, get_pretraining_datasets  # noqa: E402
from magma_amd.train_engine import initialize  # noqa: E402
from magma_amd.train_loop import eval_step, inference_step, train_step  # noqa: E402
from magma_amd.utils import (configure_param_groups, cycle, init_distributed, load_model, parse_args, print_main,  # noqa: E402
                             save_model)

if __name__ == "__main__":
    args = parse_args()
    local_rank, rank, world = init_distributed()
    model = Magma(args.config, device=torch.device("cuda", max(local_rank, 0)))
    tokenizer, config, transforms = model.tokenizer, model.config, model.transforms
    if args.synthetic_steps is not None:
        config.train_steps = args.synthetic_steps
    if args.grad_accum is not None:
        config.gradient_accumulation_steps = args.grad_accum
        config.deepspeed_config_params["gradient_accumulation_steps"] = args.grad_accum
    trainable_parameters = configure_param_groups(model, config)

    def synthetic(n, seed):
        return lambda: SyntheticImgCptDataset(n, image_size=config.image_size, seq_len=model.seq_len, eos=model.eos_token,
                                              vocab=model.eos_token, seed=seed + 1000 * rank)

    train_dataset, eval_dataset = get_pretraining_datasets(        # reference train.py:45-66
        config, tokenizer, transforms, seq_len=model.seq_len, synthetic_train=synthetic(1 << 16, 1234),
        synthetic_eval=synthetic(1 << 10, 4321), split_seed=0)
    print_main(f"Loaded train dataset with {len(train_dataset)} samples")
    print_main(f"Loaded eval dataset with {len(eval_dataset)} samples")

    model_engine, opt, train_loader, lr_scheduler = initialize(
        model=model, config=config, model_parameters=trainable_parameters, training_data=None)
    train_loader = model_engine.deepspeed_io(train_dataset, batch_size=args.micro_batch)
    eval_loader = cycle(model_engine.deepspeed_io(eval_dataset, batch_size=args.micro_batch))
    train_loader = cycle(train_loader)

    global_step = 0
    if config.load:
        previous = load_model(model_engine, config.load, load_optimizer_states=config.load_optimizer,
                              load_lr_scheduler_states=config.load_optimizer)
        if config.load_optimizer:
            global_step = previous
    model_engine.train()
    while global_step < config.train_steps:
        loss = train_step(config, train_loader, model_engine)
        global_step += 1
        if global_step % config.log_every == 0:
            print_main(f"training... Step: {global_step} Loss: {loss} lr: {lr_scheduler.get_lr()}")
        if global_step % config.eval_every == 0:
            model_engine.eval()
            with torch.no_grad():
                eval_loss = eval_step(config, eval_loader, model_engine)
                print_main(f"evaluating... Step: {global_step} Eval Loss: {eval_loss}")
                _, caption = inference_step(config, eval_loader, model_engine)
                print_main(caption)
            model_engine.train()
        if global_step % config.save_every == 0 and config.save is not None:
            save_model(model_engine, config.save, global_step)
            print_main(f"saving model at step {global_step}")
    if config.save is not None:
        save_model(model_engine, config.save, global_step)
        print_main(f"saving model at end of training (step {global_step})")
