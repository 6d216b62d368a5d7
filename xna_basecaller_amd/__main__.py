"""`python -m xna_basecaller_amd basecaller MODEL_DIR READS_DIR ...` == `bonito basecaller ...`, and `... evaluate MODEL_DIR
--directory CTC_DATA` == `bonito evaluate ...` (bonito/__init__.py:10-33)."""
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

from . import __version__
from .cli import basecaller, evaluate


def main():
    parser = ArgumentParser("bonito", formatter_class=ArgumentDefaultsHelpFormatter)
    parser.add_argument("-v", "--version", action="version", version="%(prog)s {}".format(__version__))
    sub = parser.add_subparsers(title="subcommands", description="valid commands", help="additional help",
                                dest="command")
    sub.required = True
    p = sub.add_parser("basecaller", parents=[basecaller.argparser()])
    p.set_defaults(func=basecaller.main)
    p = sub.add_parser("evaluate", parents=[evaluate.argparser()])
    p.set_defaults(func=evaluate.main)
    args = parser.parse_args()
    args.func(args)


if __name__ == "__main__":
    main()
