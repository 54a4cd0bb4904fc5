// Column layout of the fp64 result table, per calculation kind.
// Mirrored by soap_amd/layout.py (tests/test_abi.py checks the two agree).
#pragma once

// ---- SOAP_CALC_SUBHALO (BoundSubhalo) -----------------------------------------
enum {
    SH_Mtot = 0, SH_Mgas, SH_Mdm, SH_Mstar, SH_Mbh,
    SH_Ngas, SH_Ndm, SH_Nstar, SH_Nbh,
    SH_com = 9, SH_vcom = 12, SH_Ldm = 15,
    SH_Vmax_unsoft = 18, SH_Vmax_soft, SH_R_vmax_unsoft, SH_spin,
    SH_HMR_tot = 22, SH_HMR_dm,
    SH_veldisp_dm = 24,
    SH_EncloseRadius = 30,
    // gas / star / baryon kinematics (zero in dark-matter-only chunks)
    SH_com_star = 31, SH_Lgas = 34, SH_Lstar = 37, SH_Lbaryons = 40,
    SH_veldisp_gas = 43, SH_veldisp_star = 49,
    SH_HMR_gas = 55, SH_HMR_star, SH_HMR_baryon,
    // ordered rotation: kappa_corot of gas / stars / baryons, disc-to-total fractions of gas / stars
    SH_kappa = 58, SH_DtoT = 61,
    // inertia tensors (xx yy zz xy xz yz each): total {iterative, reduced, fixed sphere, reduced fixed sphere},
    // then the same four for the dark matter, the gas and the stars
    SH_inertia = 63,
    SH_NCOLS = 159
};
// ---- SOAP_CALC_SO_* -------------------------------------------------------------
enum {
    SOC_r = 0, SOC_Mtot,
    SOC_Ngas, SOC_Ndm, SOC_Nstar, SOC_Nbh,
    SOC_com = 6, SOC_vcom = 9,
    SOC_Vmax_soft = 12, SOC_R_vmax_soft,
    SOC_Mfrac_sat = 14, SOC_Mfrac_ext,
    SOC_Mgas = 16, SOC_Mdm,
    SOC_Ldm = 18,
    SOC_Mstar = 21, SOC_Mbh,
    SOC_spin = 23, SOC_conc_unsoft, SOC_conc_soft,
    SOC_com_gas = 26, SOC_com_star = 29, SOC_vcom_gas = 32, SOC_vcom_star = 35,
    SOC_Lgas = 38, SOC_Lstar = 41, SOC_Lbaryons = 44,
    SOC_Nnu = 47, SOC_Mnu, SOC_MnuNS,   // neutrino particles inside the sphere
    SOC_flow = 50,                      // dark matter mass flow rate: inflow at 0.1 / 0.3 / 1 R_SO, outflow at the same
    SOC_DtoT = 56,                      // disc-to-total fractions of gas / stars
    SOC_inertia = 58,                   // sixteen inertia tensors, as in the bound subhalo block
    SOC_NCOLS = 154
};
enum { INERTIA_NCOLS = 96
};
// ---- SOAP_CALC_APERTURE_* ---------------------------------------------------------
enum {
    AP_Mtot = 0, AP_Mgas, AP_Mdm, AP_Mstar, AP_Mbh,
    AP_Ngas, AP_Ndm, AP_Nstar, AP_Nbh,
    AP_com = 9, AP_vcom = 12, AP_Ldm = 15,
    AP_veldisp_dm = 18,
    AP_HMR_dm = 24, AP_Vmax_soft, AP_R_vmax_soft,
    AP_com_star = 27, AP_vcom_star = 30, AP_Lgas = 33, AP_Lstar = 36, AP_Lbaryons = 39,
    AP_veldisp_gas = 42, AP_veldisp_star = 48,
    AP_HMR_gas = 54, AP_HMR_star, AP_HMR_baryon,
    AP_kappa = 57, AP_DtoT = 60,        // kappa_corot gas / stars / baryons, disc-to-total gas / stars
    AP_NCOLS = 62
};
// ---- SOAP_CALC_PROJECTED: 3 blocks (projx, projy, projz) of PJ_NCOLS ---------------
enum {
    PJ_Mtot = 0, PJ_Mgas, PJ_Mdm, PJ_Mstar, PJ_Mbh,
    PJ_Ngas, PJ_Ndm, PJ_Nstar, PJ_Nbh,
    PJ_com = 9, PJ_vcom = 12,
    PJ_HMR_dm = 15, PJ_veldisp_dm,
    PJ_com_star = 17, PJ_veldisp_gas = 20, PJ_veldisp_star,
    PJ_HMR_gas = 22, PJ_HMR_star, PJ_HMR_baryon,
    PJ_NCOLS = 25
};
// 2D inertia tensors of a projected aperture, only with SOAP_CALC_FLAG_INERTIA: after the three projection blocks,
// per projection 36 columns = {total, gas, stars} x {iterative, reduced, fixed circle, reduced fixed circle} x (11, 22, 12)
enum { PJ_INERTIA_PER_AXIS = 36, PJ_INERTIA_NCOLS = 108 };
