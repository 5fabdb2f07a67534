/* ref_legacy_rxstream.h -- TEST INFRASTRUCTURE ONLY (see ref_legacy_pre.h): included by the scratch copy of bb/mod/fetchdt.h, after SignalBlock is known. */
#pragma once
static inline HRESULT SoraRadioReadRxStream(struct _SORA_RADIO_RX_STREAM* s, FLAG* touched, SignalBlock& block)
{
    if (s->pos >= s->nblocks) return E_FETCH_SIGNAL_HW_TIMEOUT;
    memcpy(&block, (const char*)s->base + s->pos * 112, 112);
    s->pos++; *touched = 1;
    return S_OK;
}
