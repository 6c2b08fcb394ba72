"""Command line front end, `python -m muse_psfr_amd.cli` (mirrors the reference's `muse-psfr`
script, cli.py:13-122: same options, messages, three-wavelength summary and log-file format;
the reconstruction itself runs on the GPU through muse_psfr_amd.compute_psf_from_sparta)."""
import argparse
import io
import logging
import sys

from . import __version__
from . import _minifits
from .psfrec import _astropy, compute_psf_from_sparta, create_sparta_table

logger = logging.getLogger(__name__)
RULE = '-' * 68


def _setup_logging(verbose):
    """The reference routes INFO records of its package logger to stdout as '[LEVEL] message'
    (muse_psfr/__init__.py:1-14); do the same once for this package."""
    pkg = logging.getLogger('muse_psfr_amd')
    if not pkg.handlers:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter('[%(levelname)s] %(message)s'))
        pkg.addHandler(h)
    level = logging.DEBUG if verbose else logging.INFO
    pkg.setLevel(level)
    for h in pkg.handlers:
        h.setLevel(level)


def _primary_header(path):
    fits, _ = _astropy()
    if fits is not None:
        return fits.getheader(path)
    return _minifits.open(path)[0].header


def _summary(header_line, seeing, gl, l0, lbda, fwhm, beta, color):
    out = io.StringIO()
    if header_line:
        out.write(header_line + '\n')
    out.write(RULE + '\n')
    out.write('Sparta Seeing: %.2f arcsec GL: %.2f L0:%.2f m\n' % (seeing, gl, l0))
    rows = (('LBDA', '%.0f', lbda), ('FWHM', '%.2f', fwhm), ('BETA', '%.2f', beta))
    if color:
        from colorama import Back, Fore, Style
        on = Back.BLACK + Style.BRIGHT + Fore.WHITE
        off = Fore.RESET + Style.NORMAL + Back.RESET
        for name, fmt, vals in rows:
            cells = ' '.join(c + fmt % v for c, v in zip((Fore.BLUE, Fore.GREEN, Fore.RED), vals))
            out.write('%s%s %s%s\n' % (on, name, cells, off))
        out.write(Style.RESET_ALL)
    else:
        for name, fmt, vals in rows:
            out.write('%s %s\n' % (name, ' '.join(fmt % v for v in vals)))
    out.write(RULE + '\n')
    return out.getvalue()


def main(args=None):
    parser = argparse.ArgumentParser(description='MUSE-PSFR (MI355X) version %s' % __version__)
    add = parser.add_argument
    add('raw', nargs='?', help='observation raw file name')
    add('--values', help='values of seeing, GL, L0, to use instead of the raw file, '
        'comma-separated')
    add('--logfile', default='muse_psfr.log', help='name of log file')
    add('-o', '--outfile', help='name of a FITS file in which the results are saved: table with '
        'individual and mean Moffat fits, and mean reconstructed PSF')
    add('--njobs', default=-1, type=int, help='accepted for compatibility (rows are one GPU batch)')
    add('--verbose', '-v', action='store_true', help='verbose flag')
    add('--no-color', action='store_true', help='no color in output')
    add('--plot', action='store_true', help='plot reconstructed psf')
    add('--device', default=None, type=int, help='GPU index (default: device 0 for a single PSF; a SPARTA\n'
        'table fans out over the visible GPUs)')
    add('--version', action='version', version='%(prog)s ' + __version__)
    opt = parser.parse_args(args)

    _setup_logging(opt.verbose)
    logger.info('MUSE-PSFR version %s', __version__)

    header_line = None
    if opt.values:
        try:
            values = [float(x) for x in opt.values.split(',')]
        except ValueError:
            values = []
        if len(values) != 3:
            sys.exit('--values must contain a list of 3 comma-separated values for seeing, GL, '
                     'and L0')
        hdu = create_sparta_table(seeing=values[0], GL=values[1], L0=values[2])
        fits, _ = _astropy()
        source = (fits.HDUList([fits.PrimaryHDU(), hdu]) if fits is not None
                  else _minifits.HDUList([_minifits.PrimaryHDU(), hdu]))
    else:
        if opt.raw is None:
            sys.exit('no input file provided')
        source = opt.raw
        hdr = _primary_header(opt.raw)
        header_line = 'OB %s %s Airmass %.2f-%.2f' % (
            hdr.get('HIERARCH ESO OBS NAME', hdr.get('ESO OBS NAME')), hdr.get('DATE'),
            hdr.get('HIERARCH ESO TEL AIRM START', hdr.get('ESO TEL AIRM START', 0)) or 0,
            hdr.get('HIERARCH ESO TEL AIRM END', hdr.get('ESO TEL AIRM END', 0)) or 0)
        logger.info(header_line)

    logger.info('Computing PSF Reconstruction from Sparta data')
    res = compute_psf_from_sparta(source, lmin=500, lmax=900, nl=3, n_jobs=opt.njobs,
                                  plot=opt.plot, device=opt.device)
    if not res:
        sys.exit('No results')
    data = res['FIT_MEAN'].data
    hdr = res['FIT_MEAN'].header
    color = not opt.no_color
    if color:
        try:
            import colorama  # noqa: F401
        except ImportError:
            color = False
    text = _summary(header_line, hdr['SEEING'], hdr['GL'], hdr['L0'], data['lbda'] * 10,
                    data['fwhm'][:, 0], data['n'], color)
    for line in text.splitlines():
        logger.info(line)
    if opt.logfile is not None:
        with open(opt.logfile, 'a') as fd:
            fd.write('\nFile: {}\n'.format(opt.raw))
            fd.write(text)
        logger.info('Results saved to %s' % opt.logfile)
    if opt.outfile is not None:
        res.writeto(opt.outfile, overwrite=True)
        logger.info('FITS file saved to %s' % opt.outfile)


if __name__ == '__main__':
    main()
