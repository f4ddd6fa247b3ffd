"""`peakachu pool`: print the loop calls (cluster representatives) of a scored-pixel
bedpe.  Mirror of peakachu/call_loops.py:3-26: same arguments (`resolution`, `infile`,
`outfile`, `threshold`), same 8-column output in the same order."""
from . import peakacluster


def main(args):
    res = args.resolution
    clusters, score_pool = peakacluster.parse_peakachu(args.infile, args.threshold, res)
    with open(args.outfile, "w") as out:
        for chrom, pixels in clusters.items():
            scored = score_pool[chrom]
            for pix in pixels:
                if pix not in scored:
                    continue
                prob, signal = scored[pix]
                a, b = pix[0] * res, pix[1] * res
                out.write("\t".join([chrom, str(a), str(a + res), chrom, str(b), str(b + res),
                                     str(prob), str(signal)]) + "\n")
