#!/usr/bin/env python
"""Command-line entry point of the controller (reference: app/main.py:9-48; same flags, plus
``--engine`` / ``--gpus`` / ``--encrypt`` / ``--nodes`` ... to generate and run a scenario without the
web UI)."""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from p2pfl_amd import __version__  # noqa: E402
from p2pfl_amd.controller import Controller  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))

argparser = argparse.ArgumentParser(description="Controller of the p2pfl_amd platform", add_help=False)
argparser.add_argument("--cloud", dest="cloud", action="store_true", default=False, help="Deploy in a cloud environment")
argparser.add_argument("-f", "--federation", dest="federation", default="DFL", help="Federation architecture: CFL, DFL, or SDFL (default: DFL)")
argparser.add_argument("-t", "--topology", dest="topology", default="fully", help="Topology: fully, ring, random, or star (default: fully)")
argparser.add_argument("-w", "--webserver", dest="webserver", action="store_true", default=False, help="Start the web server")
argparser.add_argument("-wp", "--webport", dest="webport", default=5000, help="Web server port (default: 5000)")
argparser.add_argument("-sp", "--statsport", dest="statsport", default=5100, help="Statistics (TensorBoard) port (default: 5100)")
argparser.add_argument("-s", "--simulation", action="store_false", dest="simulation", help="Disable simulation (physical devices)")
argparser.add_argument("-d", "--docker", dest="docker", action="store_true", default=False, help="Run every node in a container")
argparser.add_argument("-c", "--config", dest="config", default=os.path.join(ROOT, "config"), help="Config directory path")
argparser.add_argument("-l", "--logs", dest="logs", default=os.path.join(ROOT, "logs"), help="Logs directory path")
argparser.add_argument("-e", "--env", dest="env", default=os.path.join(ROOT, ".env"), help=".env file path")
argparser.add_argument("-v", "--version", action="version", version="%(prog)s " + __version__, help="Show version")
argparser.add_argument("-tr", "--tracking", action="store_false", dest="tracking", help="Disable tracking")
argparser.add_argument("-p", "--python", dest="python", default=sys.executable, help="Path to the python executable")
argparser.add_argument("-a", "--about", action="version", version="p2pfl_amd: MI355X-native DFL runtime", help="About")
argparser.add_argument("-h", "--help", action="help", default=argparse.SUPPRESS, help="Show help")
# MI355X additions
argparser.add_argument("--engine", default="auto", choices=["auto", "batched", "process"], help="batched: one process per GPU hosting all nodes; process: one OS process per node")
argparser.add_argument("--gpus", type=int, default=None, help="GPUs (ranks) for the batched engine")
argparser.add_argument("--control-plane", dest="control_plane", action="store_true", help="batched engine: serve the node protocol (heartbeats, roles, START/STOP_LEARNING, status reports) for the co-resident nodes")
argparser.add_argument("--gateway", action="append", default=[], metavar="IDX[:PORT]", help="batched engine: TCP endpoint of virtual node IDX for reference-protocol peers (repeatable)")
argparser.add_argument("--wait-start", dest="wait_start", type=float, default=None, help="with --gateway: wait this many seconds for START_LEARNING from an external node")
argparser.add_argument("--encrypt", action="store_true", help="encrypt model exchange / peer links")
argparser.add_argument("--scenario-name", dest="scenario_name", default=None)
argparser.add_argument("--nodes", type=int, default=0, help="generate an N-node scenario into --config before starting")
argparser.add_argument("--dataset", default="MNIST")
argparser.add_argument("--model", default="MLP")
argparser.add_argument("--rounds", type=int, default=3)
argparser.add_argument("--epochs", type=int, default=3)
argparser.add_argument("--accelerator", default="auto")
argparser.add_argument("--wait", action="store_true", help="launch the scenario, wait for it and exit (no idle loop)")


def main(argv=None):
    args = argparser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    if args.nodes:
        if not args.scenario_name:
            from datetime import datetime
            args.scenario_name = f'fedstellar_{args.federation}_{datetime.now().strftime("%d_%m_%Y_%H_%M_%S")}'
        Controller.generate_participants(os.path.join(args.config, args.scenario_name), args.nodes, args.federation,
                                         args.dataset, args.model, args.rounds, args.epochs,
                                         accelerator=args.accelerator,
                                         controller=f"127.0.0.1:{args.webport}" if args.webserver else "")
    os.makedirs(args.logs, exist_ok=True)
    controller = Controller(args)
    if args.wait:
        controller.load_configurations_and_start_nodes()
        codes = controller.wait()
        sys.exit(0 if all(c == 0 for c in codes) else 1)
    controller.start()


if __name__ == "__main__":
    main()
