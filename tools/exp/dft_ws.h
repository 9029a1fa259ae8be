// Interface of the wave-specialised DFT pass experiment (dft_ws.hip): the arguments of the product kernel (dft_rx3.h) plus
// the fused spectral mix as per-tile tables.
#pragma once
#include "../../surfh_amd/csrc/dft_rx3.h"
struct DftWsArgs : DftRx3Args {
    // mixtab[kb][2 k + c] = (mhat[t][c][k][kb])_t (launch_dft_ws_mix_table, mix_rows = dft_ws_mix_rows(Kn, KP) rows k, zero
    // beyond Kn) and tplT[l] = (tpl[t][l])_t
    const float4 *mixtab = nullptr, *tplT = nullptr;
    int mix_rows = 0;
};
bool dft_ws_can(const DftWsArgs &g);
int launch_dft_ws(hipStream_t stream, const DftWsArgs &g);
int dft_ws_mix_rows(int Kn, int KP);
int launch_dft_ws_mix_table(hipStream_t stream, const float *mhat, float *mixtab, int T, int Kn, int nkb, long PL, long KBP, int mix_rows);
