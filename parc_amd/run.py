"""Launcher with the reference's run.py flags (run.py:95-164): --env_config --agent_config --mode train|test|record
--num_envs --device --max_samples --out_model_file --int_output_dir --log_file --model_file --rand_seed --test_episodes.
One process per GPU: under torchrun (RANK/WORLD_SIZE set) every rank takes its own device and RCCL carries the gradient
all-reduce; --num_workers spawns local ranks like the reference does."""
import os
import sys
import time

import numpy as np
import torch

from .envs import env_builder
from .learning import agent_builder
from .util import arg_parser, mp_util, util


def load_args(argv):
    args = arg_parser.ArgParser()
    args.load_args(argv[1:])
    arg_file = args.parse_string("arg_file", "")
    if arg_file != "":
        assert args.load_file(arg_file), "Failed to load args from: " + arg_file
    return args


def run(rank, num_procs, master_port, args):
    mode = args.parse_string("mode", "train")
    num_envs = args.parse_int("num_envs", 1)
    device = args.parse_string("device", "cuda:0")
    mp_util.init(rank, num_procs, device, master_port)       # maps a bare cuda / cuda:0 to this rank's GPU (mp_util.rank_device)
    device = mp_util.get_device()
    seed = args.parse_int("rand_seed") if args.has_key("rand_seed") else int(time.time() * 256) % (2 ** 31)
    util.set_rand_seed(seed + 41 * mp_util.get_proc_rank())            # run.py:90
    out_model_file = args.parse_string("out_model_file", "output/model.pt")
    int_output_dir = args.parse_string("int_output_dir", "")
    if mp_util.is_root_proc():
        for d in (os.path.dirname(out_model_file), int_output_dir):
            if d:
                os.makedirs(d, exist_ok=True)
    env = env_builder.build_env(args.parse_string("env_config"), num_envs, device, args.parse_bool("visualize", False))
    agent = agent_builder.build_agent(args.parse_string("agent_config"), env, device)
    model_file = args.parse_string("model_file", "")
    if model_file != "":
        agent.load(model_file)
    if mode == "train":
        agent.train_model(max_samples=args.parse_int("max_samples", np.iinfo(np.int64).max), out_model_file=out_model_file,
                          int_output_dir=int_output_dir, log_file=args.parse_string("log_file", "output/log.txt"),
                          logger_type=args.parse_string("logger", "tb"))
    elif mode == "test":
        res = agent.test_model(num_episodes=args.parse_int("test_episodes", 16))
        print("Mean Return: {}\nMean Episode Length: {}\nEpisodes: {}".format(res["mean_return"], res["mean_ep_len"], res["num_eps"]))
    elif mode == "record":
        agent.record_motions()
    else:
        raise AssertionError("Unsupported mode: {}".format(mode))


def main(argv):
    args = load_args(argv)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:            # torchrun: one process per GPU already
        run(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), None, args)
        return
    num_workers = args.parse_int("num_workers", 1)
    master_port = args.parse_int("master_port", None) or np.random.randint(6000, 7000)
    procs = []
    if num_workers > 1:
        torch.multiprocessing.set_start_method("spawn", force=True)
        for r in range(1, num_workers):
            p = torch.multiprocessing.Process(target=run, args=[r, num_workers, master_port, args])
            p.start()
            procs.append(p)
    run(0, num_workers, master_port, args)
    for p in procs:
        p.join()


if __name__ == "__main__":
    main(sys.argv)
