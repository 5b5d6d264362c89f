"""
Command line: the ``eigenvals`` command of the reference's ``tbmodels`` CLI
(`src/tbmodels/_cli.py:227-262`) -- the step either side of the hot path: model + k-points from HDF5
files, ONE batched ``Model.eigenval`` call on the GPU, eigenvalues back to an HDF5 file.

    python -m tbmodels_amd eigenvals -i model.hdf5 -k kpoints.hdf5 -o eigenvals.hdf5 [-v]

Options, defaults and file formats are the reference's; the other ``tbmodels`` commands (``parse``,
``symmetrize``, ``slice``) are outside the hot path and are not provided.
"""

import argparse
import sys

from . import io

__all__ = ("main",)


def _eigenvals(args):
    def echo(message):
        if args.verbose:
            print(message)

    echo("Reading initial model from file '{}' ...".format(args.input))
    model = io.load(args.input)
    echo("Reading kpoints from file '{}' ...".format(args.kpoints))
    kpts = io.load(args.kpoints)
    if isinstance(kpts, io.EigenvalsData):
        kpts = kpts.kpoints
    echo("Calculating energy eigenvalues ...")
    eigenvalues = io.EigenvalsData.from_eigenval_function(
        kpoints=kpts, eigenval_function=model.eigenval_array, listable=True  # one batched call, no list of rows
    )
    echo("Writing kpoints and energy eigenvalues to file '{}' ...".format(args.output))
    io.save(eigenvalues, args.output)
    echo("Done!")
    return 0


def main(argv=None):
    parser = argparse.ArgumentParser(prog="tbmodels_amd", description="k-space evaluation of tight-binding models.")
    commands = parser.add_subparsers(dest="command", required=True)
    eig = commands.add_parser(
        "eigenvals",
        help="Calculate energy eigenvalues.",
        description="Calculate the energy eigenvalues for a given set of k-points (in reduced coordinates). "
        "The input and output is given in an HDF5 file.",
    )
    eig.add_argument("-i", "--input", default="model.hdf5", help="File containing the input model (in HDF5 format).")
    eig.add_argument(
        "-k",
        "--kpoints",
        default="kpoints.hdf5",
        help="File containing the k-points for which the eigenvalues are evaluated.",
    )
    eig.add_argument("-o", "--output", default="eigenvals.hdf5", help="Output file for the energy eigenvalues.")
    eig.add_argument("-v", "--verbose", action="store_true", help="Enable verbose output.")
    eig.set_defaults(run=_eigenvals)
    args = parser.parse_args(argv)
    return args.run(args)


if __name__ == "__main__":
    sys.exit(main())
