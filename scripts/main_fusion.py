"""MRS fusion driver on MI355X -- same command line, same result files as the reference's driver
(scripts/main_fusion.py:160-274 of sidiso/surfh), with the HIP operator and the device-resident CG behind it.

    python scripts/main_fusion.py -fd <fusion_dir> -np 501 -hp 5e3 -ni 50 -nt 4 -m lcg
    python scripts/main_fusion.py --synthetic config2 -hp 5e3 -ni 50         # no input files needed

Inputs under ``fusion_dir`` (reference layout, main_fusion.py:65-75): ``Templates/`` (wavelength axis + NMF templates,
.npy), ``PSF/`` (PSF stack, .npy), ``Filtered_slices/`` (one FITS file per band and pointing) -> results in
``Results/<method>_MC_<channels>_MO_4_Temp_<T>_nit_<niter>_mu_<mu>_SD_<scale>/``: ``res_x.npy`` (abundance maps),
``res_cube.npy`` (``mapsToCube`` of them), ``criterion.npy`` (criterion trace, fusion_CT.py:163-175,242-265).

The FITS reader needs astropy (FITS I/O is outside the hot path and not rebuilt here); when it is not importable the
same arrays may be given as ``Filtered_slices/<band>_<k>.npz`` with fields ``data`` (raveled ``[Ldet, S, a_out]`` as in
the FITS primary HDU), ``PA_V3``, ``TARG_RA``, ``TARG_DEC``.  ``--synthetic`` builds one of the benchmark problems
(surfh_amd/synth.py), simulates the slit data with the operator and runs the same reconstruction.
"""
import logging as log
import os
import pathlib
import sys

import click
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfh_amd import instru, synth                                   # noqa: E402
from surfh_amd.fusion import QuadCriterion_MRS                        # noqa: E402
from surfh_amd.models import spectroSigRLSCT                          # noqa: E402

LIST_CHAN = ['1a', '1b', '1c', '2a', '2b', '2c', '3a', '3b', '3c', '4a', '4b', '4c']

def initialize_parameters(fusion_dir_path):
    """Paths and the spatial step (main_fusion.py:65-78)."""
    paths = {
        'psf_dir': os.path.join(fusion_dir_path, 'PSF/'),
        'template_dir': os.path.join(fusion_dir_path, 'Templates/'),
        'save_filter_corrected_dir': os.path.join(fusion_dir_path, 'Filtered_slices/'),
        'result_path': os.path.join(fusion_dir_path, 'Results/'),
        'mask_path': os.path.join(fusion_dir_path, 'Masks/')
    }
    step = 0.025  # arcsec
    return paths, step, step / 3600.0


def load_simulation_data(paths, step, step_angle, npix, n_templates):
    """Axes, templates and OTF (main_fusion.py:80-101)."""
    imshape = (npix, npix)
    origin_alpha_axis = synth.axes(npix, step_angle)
    origin_beta_axis = synth.axes(npix, step_angle)
    if n_templates not in (4, 6):
        raise NameError("No corresponding Templates name")
    tag = f'orion_1ABC_2ABC_3ABC_4ABC_{n_templates}_templates_SS4.npy'
    wavel_axis = np.load(os.path.join(paths['template_dir'], 'wavel_axis_' + tag))
    templates = np.load(os.path.join(paths['template_dir'], 'nmf_' + tag))
    spsf = np.load(os.path.join(paths['psf_dir'], 'psfs_pixscale0.025_npix_501_fov12.525_chan_1ABC_2ABC_3ABC_4ABC_SS4.npy'))
    sotf = synth.ir2fr(spsf, imshape)
    templates = templates / 10e3
    return origin_alpha_axis, origin_beta_axis, wavel_axis, templates, sotf


def load_data(list_chan, save_filter_corrected_dir):
    """Slit data, pointing targets and roll angle per band (main_fusion.py:30-63)."""
    data_dict = {'data': {c: [] for c in list_chan}, 'target': {c: [] for c in list_chan},
                 'rotation': {c: 0. for c in list_chan}}
    try:
        from astropy.io import fits
    except ImportError:
        fits = None
    for file in sorted(os.listdir(save_filter_corrected_dir)):
        for chan in list_chan:
            if chan not in file:
                continue
            n_slit, n_det = synth.BANDS[chan][0], synth.BANDS[chan][6][2]
            full = os.path.join(save_filter_corrected_dir, file)
            if file.endswith('.npz'):
                z = np.load(full)
                data, pa, ra, dec = z['data'], float(z['PA_V3']), float(z['TARG_RA']), float(z['TARG_DEC'])
            elif fits is not None:
                with fits.open(full) as hdul:
                    h = hdul[0].header
                    data, pa, ra, dec = hdul[0].data, h['PA_V3'], h['TARG_RA'], h['TARG_DEC']
            else:
                raise RuntimeError(f"{file}: reading FITS needs astropy, which is not installed; "
                                   "provide <band>_<k>.npz files instead (see the module docstring)")
            # the primary HDU is [Ldet, S, a_out] raveled (a_out = 19 / 24 / 24 / 27 for channels 1-4); the model's
            # channel output is [S, Ldet, a_out] per pointing
            ndata = np.asarray(data).reshape(n_det, n_slit, -1).transpose(1, 0, 2)
            data_dict['data'][chan].append(ndata)
            data_dict['target'][chan].append((ra, dec))
            data_dict['rotation'][chan] = pa
    return data_dict


def create_instruments(data_dict, list_chan=LIST_CHAN):
    """One IFU per band with the constants of main_fusion.py:107-136 (held in surfh_amd.synth.BANDS)."""
    return {chan: synth.band_ifu(chan, angle=-data_dict['rotation'][chan]) for chan in list_chan}


def create_model(sotf, templates, origin_alpha_axis, origin_beta_axis, wavel_axis, instruments, step_angle, data_dict,
                 device=0):
    """main_fusion.py:138-158: pointings from the FITS targets, axes recentred on the third 2A pointing."""
    main_pointing = instru.Coord(0, 0)
    pointings = []
    for chan in instruments.keys():
        pointing_chan = [main_pointing + instru.Coord(ra, dec) for ra, dec in data_dict['target'][chan]]
        pointings.append(instru.CoordList(pointing_chan).pix(step_angle))
    ref = data_dict['target']['2a'][2] if len(data_dict['target'].get('2a', [])) > 2 else (0.0, 0.0)
    return spectroSigRLSCT(sotf=sotf, templates=templates, alpha_axis=origin_alpha_axis + ref[0],
                           beta_axis=origin_beta_axis + ref[1], wavelength_axis=wavel_axis,
                           instrs=list(instruments.values()), step_degree=step_angle, pointings=pointings, device=device)


def result_dir_name(method, n_channels, n_templates, niter, hyper_parameter, scale_data):
    """main_fusion.py:182."""
    return f'{method}_MC_{n_channels}_MO_4_Temp_{n_templates}_nit_{str(niter)}_mu_{str("{:.2e}".format(hyper_parameter))}_SD_{scale_data}/'


def reconstruction_method(spectro_model, ndata, templates, result_path, hyper_parameter, niter, method, scale_data,
                          checkpoint_every=0, resume=None):
    """main_fusion.py:162-206: regularised least squares by CG, then the three result files.  Not in the reference:
    `checkpoint_every` > 0 writes the iterate to checkpoint.npz in the result directory every that many iterations,
    `resume` (such a file) warm-starts from it and runs the iterations that are left."""
    value_init = 0
    path = pathlib.Path(result_path) / result_dir_name(method, len(spectro_model.instrs), templates.shape[0], niter,
                                                       hyper_parameter, scale_data)
    path.mkdir(parents=True, exist_ok=True)
    crit = QuadCriterion_MRS(mu_spectro=1, y_spectro=np.copy(ndata), model_spectro=spectro_model,
                             mu_reg=hyper_parameter, printing=True, gradient="separated")
    if resume:
        from surfh_amd.fusion import load_checkpoint
        x_saved, it_done, _ = load_checkpoint(resume)
        value_init = np.asarray(x_saved, dtype=np.float64).reshape(crit.shape_of_output)
        print(f"Resuming from {resume}: {it_done} iterations done, {max(niter - it_done, 0)} to go")
        niter = max(niter - it_done, 0)
    ck = (path / 'checkpoint.npz', checkpoint_every) if checkpoint_every and checkpoint_every > 0 else None
    res = crit.run_method(method, niter, perf_crit=1, calc_crit=True, value_init=value_init, checkpoint=ck)
    y_cube = spectro_model.mapsToCube(res.x)
    print(f"Results save in {path}")
    np.save(path / 'res_x.npy', res.x)
    np.save(path / 'res_cube.npy', y_cube)
    np.save(path / 'criterion.npy', crit.L_crit_val)
    return res, path


def synthetic_problem(name, npix):
    """One of the benchmark problems + simulated data  y = A maps + noise  (SURVEY.md 8d)."""
    if name == 'small':
        prob = synth.problem(['2a'], 256, (7.41, 8.87), n_pix=npix)
    elif name in ('config2', 'config3', 'config4'):
        prob = getattr(synth, name)(n_pix=npix)
    else:
        raise click.BadParameter(f"unknown synthetic problem {name!r} (small, config2, config3, config4)")
    return prob


@click.command()
@click.option('-fd', '--fusion_dir', default='/home/nmonnier/Data/JWST/Orion_bar/Fusion/', type=str, help='Fusion directory')
@click.option('-np', '--npix', default=501, type=int, help='Number of pixels')
@click.option('-hp', '--hyper_parameter', default=1., type=float, help='Hyperparameter value')
@click.option('-ni', '--niter', default=5, type=int, help='Number of iteration.')
@click.option('-nt', '--n_templates', default=4, type=int, help='Number of Templates.')
@click.option('-sd', '--scale_data', default=False, type=bool, help='Scale data from Jy  to Jy/str.')
@click.option('-m', '--method', default='lcg', type=str, help='Method used (default = lcg).')
@click.option('-v', '--verbose', default=True, type=bool, help='Verbose.')
@click.option('--synthetic', default=None, type=str,
              help='Run on a synthetic benchmark problem (small, config2, config3, config4) instead of fusion_dir inputs; '
                   'results go to <fusion_dir>/Results/.')
@click.option('--device', default=0, type=int, help='GPU index.')
@click.option('--checkpoint_every', default=0, type=int, help='Write the iterate to checkpoint.npz every that many iterations (0: never).')
@click.option('--resume', default=None, type=str, help='checkpoint.npz of an interrupted run to warm-start from.')
def main(fusion_dir, npix, hyper_parameter, niter, n_templates, scale_data, method, verbose, synthetic, device, checkpoint_every=0,
         resume=None):
    print('options:', dict(fusion_dir=fusion_dir, npix=npix, hyper_parameter=hyper_parameter, niter=niter,
                           n_templates=n_templates, scale_data=scale_data, method=method, synthetic=synthetic, device=device))
    if verbose:
        log.basicConfig(format="%(levelname)s: %(message)s", level=log.INFO)

    log.info('Initialize basic path parameters')
    paths, step, step_angle = initialize_parameters(fusion_dir)

    if synthetic:
        log.info(f'Build the synthetic problem {synthetic}')
        prob = synthetic_problem(synthetic, npix)
        templates = prob['templates']
        model = spectroSigRLSCT(prob['sotf'], templates, prob['alpha_axis'], prob['beta_axis'], prob['wavel'],
                                prob['ifus'], prob['step_deg'], prob['pointings'], device=device)
        y = model.forward(prob['maps'])
        ndata = y + np.random.default_rng(1).standard_normal(y.shape) * 1e-2 * np.sqrt(np.mean(y ** 2))
    else:
        log.info('Load simulation data')
        origin_alpha_axis, origin_beta_axis, wavel_axis, templates, sotf = load_simulation_data(paths, step, step_angle, npix, n_templates)
        log.info('Load MRS data')
        data_dict = load_data(LIST_CHAN, paths["save_filter_corrected_dir"])
        log.info('Create instruments and spectro models')
        instruments = create_instruments(data_dict)
        model = create_model(sotf, templates, origin_alpha_axis, origin_beta_axis, wavel_axis, instruments, step_angle,
                             data_dict, device=device)
        ndata = np.concatenate([np.array(data_dict['data'][chan]).ravel() for chan in LIST_CHAN])

    if scale_data:
        log.info('Data scaling enable')
        ndata = model.real_data_janskySR_to_jansky(ndata)

    log.info(f'Start {method} algorithm')
    reconstruction_method(model, ndata, templates, paths["result_path"], hyper_parameter, niter, method, scale_data,
                          checkpoint_every=checkpoint_every, resume=resume)
    model.close()


if __name__ == '__main__':
    main()
