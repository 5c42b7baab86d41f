// decode_fields.hip -- SURVEY.md 8 (f)4: the rest of decodeModesMessage for the accepted set, one lane per message.
//
// What mode_s.c:557-716 (header fields by downlink format), decodeExtendedSquitter (mode_s.c:736-1474, DF17 / DF18) and
// decodeModeAMessage (mode_ac.c:165-199) leave in struct modesMessage (readsb.h:421-635), written as the 416-byte record
// rsb_decoded of include/readsb_amd.h -- the reference's own x86-64 layout, so a caller copies it over its struct.
//
// Shape of the kernel: a message is 14 bytes in and 416 bytes out, most of them zero, so the work is byte traffic, not
// arithmetic.  A workgroup of 128 lanes zeroes 128 records in LDS (odd stride of 105 dwords: the lanes' writes fall on
// different banks), every lane sets the members its message defines, and the workgroup stores the 53 KB as one contiguous
// run of dwords (whole 256-byte rows per wavefront instruction).  Field extraction works on two registers per message: the
// first 32 bits of the frame and the 56-bit ME / MB field as one 64-bit word, so every getbits(..., first, last) of the
// reference (mode_s.h:58-115, bits numbered from 1, MSB first) is one shift and one mask.
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/readsb_amd.h"
#include "kernels.hpp"

namespace rsb {
namespace {

constexpr int kLanes = 128;
constexpr int kRecWords = RSB_DECODED_BYTES / 4;  // 104
constexpr int kStride = kRecWords + 1;            // dwords between the lanes' records in LDS

// readsb.h:131-246
enum : int { kSrcModeAC = 1, kSrcModeS = 3, kSrcModeSChecked = 4, kSrcTisb = 5, kSrcAdsr = 6, kSrcAdsb = 7 };
enum : int { kAddrAdsbIcao = 0, kAddrAdsbIcaoNt = 1, kAddrAdsrIcao = 2, kAddrTisbIcao = 3, kAddrAdsbOther = 4, kAddrAdsrOther = 5, kAddrTisbTrackfile = 6,
             kAddrTisbOther = 7, kAddrModeA = 8, kAddrUnknown = 9 };
enum : int { kAgInvalid = 0, kAgGround = 1, kAgAirborne = 2, kAgUncertain = 3 };
enum : int { kHeadingGroundTrack = 1, kHeadingTrue = 2, kHeadingMagnetic = 3, kHeadingMagneticOrTrue = 4, kHeadingTrackOrHeading = 5 };
enum : int { kSilUnknown = 1, kSilPerSample = 2, kSilPerHour = 3 };
enum : int { kNavAltAircraft = 2, kNavAltMcp = 3, kNavAltFms = 4 };
enum : uint32_t { kNavAutopilot = 1, kNavVnav = 2, kNavAltHold = 4, kNavApproach = 8, kNavLnav = 16, kNavTcas = 32 };
constexpr int kInvalidAltitude = -9999;       // readsb.h:128
constexpr uint32_t kNonIcao = 1u << 24;       // readsb.h:248

// the one-bit members of readsb.h:466-492 in declaration order = bit number from byte 195 on
enum : int { fAltBaro = 0, fAltGeom, fTrack, fTrackRate, fHeading, fRoll, fGs, fIas, fTas, fMach, fBaroRate, fGeomRate, fSquawk, fCallsign, fCpr, fCprOdd,
             fCprDecoded, fCprRelative, fCategory, fGeomDelta, fFromMlat, fFromTisb, fSpiValid, fSpi, fAlertValid, fAlert, fEmergency };
// readsb.h:551-572 (one 32-bit unit at offset 348), :592-609 (at 380), :623-627 (at 400): first bit of each member
enum : int { aNicAValid = 0, aNicBValid, aNicCValid, aNicBaroValid, aNacPValid, aNacVValid, aGvaValid, aSdaValid, aNicA, aNicB, aNicC, aNicBaro, aNacP = 12,
             aNacV = 16, aSil = 19, aGva = 21, aSda = 23 };
enum : int { oValid = 0, oVersion = 1, oOmAcasRa = 4, oOmIdent, oOmAtc, oOmSaf, oCcAcas, oCcCdti, oCc1090In, oCcArv, oCcTs, oCcTc = 13, oCcUatIn = 15, oCcPoa,
             oCcB2Low, oCcLwValid };
enum : int { nHeadingValid = 0, nFmsValid, nMcpValid, nQnhValid, nModesValid };

#define OFF(member) ((int)offsetof(rsb_decoded, member))

// one lane's record in LDS
struct Rec {
    uint32_t *w;
    __device__ void u32(int off, uint32_t v) const { w[off >> 2] = v; }
    __device__ void i32(int off, int v) const { w[off >> 2] = (uint32_t)v; }
    __device__ void f32(int off, float v) const { w[off >> 2] = __float_as_uint(v); }
    __device__ void u8(int off, uint32_t v) const { ((uint8_t *)w)[off] = (uint8_t)v; }
    __device__ uint32_t get32(int off) const { return w[off >> 2]; }
    __device__ void bits(int off, int first, uint32_t v) const { w[off >> 2] |= v << first; }
    __device__ void flag(int k) const { ((uint8_t *)w)[OFF(valid_bits) + (k >> 3)] |= (uint8_t)(1u << (k & 7)); }
    __device__ void acc(int first, uint32_t v) const { bits(OFF(accuracy_bits), first, v); }
    __device__ void ops(int first, uint32_t v) const { bits(OFF(opstatus_bits), first, v); }
    __device__ void nav(int first, uint32_t v) const { bits(OFF(nav_bits), first, v); }
};

// getbits / getbit of the reference on a field held MSB-first in the low `width` bits of a word
__device__ inline uint32_t gb(uint64_t field, int width, int first, int last) {
    return (uint32_t)(field >> (width - last)) & ((1u << (last - first + 1)) - 1u);
}
struct Me {  // the 56-bit ME field
    uint64_t v;
    __device__ uint32_t bits(int first, int last) const { return gb(v, 56, first, last); }
    __device__ uint32_t bit(int k) const { return (uint32_t)(v >> (56 - k)) & 1u; }
};

// decodeID13Field (mode_s.c:101-143): C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4 -> the four octal digits A B C D as hex nibbles
__device__ uint32_t id13_to_hex(uint32_t f) {
    uint32_t h = 0;
    h |= (f & 0x1000) ? 0x0010u : 0;  // C1
    h |= (f & 0x0800) ? 0x1000u : 0;  // A1
    h |= (f & 0x0400) ? 0x0020u : 0;  // C2
    h |= (f & 0x0200) ? 0x2000u : 0;  // A2
    h |= (f & 0x0100) ? 0x0040u : 0;  // C4
    h |= (f & 0x0080) ? 0x4000u : 0;  // A4
    h |= (f & 0x0020) ? 0x0100u : 0;  // B1
    h |= (f & 0x0010) ? 0x0001u : 0;  // D1 / Q
    h |= (f & 0x0008) ? 0x0200u : 0;  // B2
    h |= (f & 0x0004) ? 0x0002u : 0;  // D2
    h |= (f & 0x0002) ? 0x0400u : 0;  // B4
    h |= (f & 0x0001) ? 0x0004u : 0;  // D4
    return h;
}

// modeAToModeC (mode_ac.c:80-86): the table is indexed by the twelve code bits only (track.h:292-295), so its entry is
// internalModeAToModeC (mode_ac.c:98-161) of the code with every other bit cleared; computed, not tabulated
__device__ int mode_a_to_c(uint32_t mode_a) {
    const uint32_t a = mode_a & 0x7777u;
    if ((a & 0x0001u) || !(a & 0x00F0u)) return kInvalidAltitude;  // D1 set, or no C bit
    uint32_t ones = ((a & 0x10) ? 7u : 0) ^ ((a & 0x20) ? 3u : 0) ^ ((a & 0x40) ? 1u : 0);
    if ((ones & 5) == 5) ones ^= 2;
    if (ones > 5) return kInvalidAltitude;
    const uint32_t fives = ((a & 0x0002) ? 0xFFu : 0) ^ ((a & 0x0004) ? 0x7Fu : 0) ^ ((a & 0x1000) ? 0x3Fu : 0) ^ ((a & 0x2000) ? 0x1Fu : 0) ^
                           ((a & 0x4000) ? 0x0Fu : 0) ^ ((a & 0x0100) ? 7u : 0) ^ ((a & 0x0200) ? 3u : 0) ^ ((a & 0x0400) ? 1u : 0);
    if (fives & 1) ones = 6 - ones;
    return (int)(fives * 5 + ones) - 13;
}

// decodeAC13Field (mode_s.c:152-179); *unit: 0 feet, 1 metres
__device__ int ac13_altitude(uint32_t f, int *unit) {
    if (f & 0x0040) {
        *unit = 1;
        return kInvalidAltitude;
    }
    *unit = 0;
    if (f & 0x0010) return (int)(((f & 0x1F80) >> 2) | ((f & 0x0020) >> 1) | (f & 0x000F)) * 25 - 1000;
    const int n = mode_a_to_c(id13_to_hex(f));
    return n < -12 ? kInvalidAltitude : 100 * n;
}

// decodeAC12Field (mode_s.c:187-208)
__device__ int ac12_altitude(uint32_t f) {
    if (f & 0x10) return (int)(((f & 0x0FE0) >> 1) | (f & 0x000F)) * 25 - 1000;
    const int n = mode_a_to_c(id13_to_hex(((f & 0x0FC0) << 1) | (f & 0x003F)));
    return n < -12 ? kInvalidAltitude : 100 * n;
}

// decodeMovementFieldV0 / V2 (mode_s.c:216-259): double expressions narrowed to float
__device__ float movement_common(uint32_t m, bool *done) {
    *done = true;
    if (m >= 125) return 0.0f;
    if (m == 124) return 180.0f;
    if (m >= 109) return (float)(100 + ((double)(m - 109) + 0.5) * 5);
    if (m >= 94) return (float)(70 + ((double)(m - 94) + 0.5) * 2);
    if (m >= 39) return (float)(15 + ((double)(m - 39) + 0.5) * 1);
    if (m >= 13) return (float)(2 + ((double)(m - 13) + 0.5) * 0.50);
    if (m >= 9) return (float)(1 + ((double)(m - 9) + 0.5) * 0.25);
    *done = false;
    return 0.0f;
}
__device__ float movement_v0(uint32_t m) {
    bool done;
    const float r = movement_common(m, &done);
    if (done) return r;
    if (m >= 2) return (float)(0.125 + ((double)(m - 2) + 0.5) * 0.125);
    return 0.0f;
}
__device__ float movement_v2(uint32_t m) {
    bool done;
    const float r = movement_common(m, &done);
    if (done) return r;
    if (m >= 3) return (float)(0.125 + ((double)(m - 3) + 0.5) * 0.875 / 6);
    if (m >= 2) return (float)(0.125 / 2);
    return 0.0f;
}

// the extended squitter's working state that later stages read back
struct Es {
    uint32_t addr;
    int addrtype, source, airground;
    uint32_t metype;
    bool check_imf;
};

// setIMF (mode_s.c:771-792)
__device__ void set_imf(Es &e) {
    e.addr |= kNonIcao;
    if (e.addrtype == kAddrAdsbIcao || e.addrtype == kAddrAdsbIcaoNt)
        e.addrtype = kAddrAdsbOther;
    else if (e.addrtype == kAddrTisbIcao)
        e.addrtype = kAddrTisbTrackfile;
    else if (e.addrtype == kAddrAdsrIcao)
        e.addrtype = kAddrAdsrOther;
}

__device__ void squawk_from(const Rec &r, uint32_t id13) {
    if (id13) {
        r.flag(fSquawk);
        r.u32(OFF(squawk), id13_to_hex(id13));
    }
}

// decodeESIdentAndCategory (mode_s.c:736-767)
__device__ void es_ident(const Rec &r, const Me &me, const Es &e) {
    const uint32_t mesub = me.bits(6, 8);
    r.u32(OFF(mesub), mesub);
    bool ok = true;
    for (int k = 0; k < 8; ++k) {
        const uint32_t six = me.bits(9 + 6 * k, 14 + 6 * k);
        const uint32_t ch = six < 32 ? 64 + six : six;  // ais_charset.c:26: '@', A-Z, "[\]^_", then ASCII 32..63 unchanged
        r.u8(OFF(callsign) + k, ch);
        ok = ok && ((ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == ' ');
    }
    if (ok) r.flag(fCallsign);
    r.u32(OFF(category), ((0x0Eu - e.metype) << 4) | mesub);
    r.flag(fCategory);
}

// decodeESAirborneVelocity (mode_s.c:794-900)
__device__ void es_velocity(const Rec &r, const Me &me, Es &e) {
    const uint32_t mesub = me.bits(6, 8);
    r.u32(OFF(mesub), mesub);
    if (mesub < 1 || mesub > 4) return;
    if (e.check_imf && me.bit(9)) set_imf(e);
    r.acc(aNacVValid, 1);
    r.acc(aNacV, me.bits(11, 13));
    if (mesub <= 2) {
        const uint32_t ew_raw = me.bits(15, 24), ns_raw = me.bits(26, 35);
        if (ew_raw && ns_raw) {
            const int scale = mesub == 2 ? 4 : 1;
            const int ew = (int)(ew_raw - 1) * (me.bit(14) ? -1 : 1) * scale, ns = (int)(ns_raw - 1) * (me.bit(25) ? -1 : 1) * scale;
            // sqrtf of the sum narrowed to float, correctly rounded: through the double root (53 bits >= 2 x 24 + 2, so rounding twice is exact)
            const float gs = (float)sqrt((double)(float)((double)(ns * ns + ew * ew) + 0.5));
            r.f32(OFF(gs_v0), gs);
            r.f32(OFF(gs_v2), gs);
            r.f32(OFF(gs_selected), gs);
            r.flag(fGs);
            if (gs > 0) {
                float track = (float)(atan2((double)ew, (double)ns) * 180.0 / 3.14159265358979323846);
                if (track < 0) track = track + 360.0f;
                r.f32(OFF(heading), track);
                r.i32(OFF(heading_type), kHeadingGroundTrack);
                r.flag(fHeading);
            }
        }
    } else {
        if (me.bit(14)) {
            r.flag(fHeading);
            r.f32(OFF(heading), (float)((double)me.bits(15, 24) * 360.0 / 1024.0));
            r.i32(OFF(heading_type), kHeadingMagneticOrTrue);
        }
        const uint32_t airspeed = me.bits(26, 35);
        if (airspeed) {
            const uint32_t speed = (airspeed - 1) * (mesub == 4 ? 4u : 1u);
            if (me.bit(25)) {
                r.flag(fTas);
                r.u32(OFF(tas), speed);
            } else {
                r.flag(fIas);
                r.u32(OFF(ias), speed);
            }
        }
    }
    const uint32_t vert = me.bits(38, 46);
    if (vert) {
        const int rate = (int)(vert - 1) * (me.bit(37) ? -64 : 64);
        if (me.bit(36)) {
            r.i32(OFF(baro_rate), rate);
            r.flag(fBaroRate);
        } else {
            r.i32(OFF(geom_rate), rate);
            r.flag(fGeomRate);
        }
    }
    const uint32_t delta = me.bits(50, 56);
    if (delta) {
        r.flag(fGeomDelta);
        r.i32(OFF(geom_delta), (int)(delta - 1) * (me.bit(49) ? -25 : 25));
    }
}

// decodeESSurfacePosition (mode_s.c:902-937)
__device__ void es_surface(const Rec &r, const Me &me, Es &e) {
    e.airground = kAgGround;
    r.flag(fCpr);
    r.i32(OFF(cpr_type), 0);  // CPR_SURFACE
    const uint32_t movement = me.bits(6, 12);
    if (movement > 0 && movement < 125) {
        r.flag(fGs);
        const float v0 = movement_v0(movement);
        r.f32(OFF(gs_selected), v0);
        r.f32(OFF(gs_v0), v0);
        r.f32(OFF(gs_v2), movement_v2(movement));
    }
    if (me.bit(13)) {
        r.flag(fHeading);
        r.f32(OFF(heading), (float)((double)me.bits(14, 20) * 360.0 / 128.0));
        r.i32(OFF(heading_type), kHeadingTrackOrHeading);
    }
    if (e.check_imf && me.bit(21)) set_imf(e);
    if (me.bit(22)) r.flag(fCprOdd);
    r.u32(OFF(cpr_lat), me.bits(23, 39));
    r.u32(OFF(cpr_lon), me.bits(40, 56));
}

// decodeESAirbornePosition (mode_s.c:939-1022); returns 1 for a position the reference counts as cpr_filtered
__device__ int es_airborne(const Rec &r, const Me &me, Es &e) {
    int filtered = 0;
    switch (me.bits(6, 7)) {
        case 0:
            r.flag(fAlertValid);
            r.flag(fSpiValid);
            break;
        case 3:
            r.flag(fAlertValid);
            r.flag(fSpiValid);
            r.flag(fSpi);
            break;
        default:  // permanent or temporary alert: nothing known about SPI
            r.flag(fAlertValid);
            r.flag(fAlert);
            break;
    }
    if (e.check_imf) {
        if (me.bit(8)) set_imf(e);
    } else {
        r.acc(aNicBValid, 1);
        r.acc(aNicB, me.bit(8));
    }
    const uint32_t ac12 = me.bits(9, 20);
    if (e.metype != 0) {
        const uint32_t lat = me.bits(23, 39), lon = me.bits(40, 56);
        r.u32(OFF(cpr_lat), lat);
        r.u32(OFF(cpr_lon), lon);
        if (ac12 == 0 && lon == 0 && (lat & 0x0fff) == 0 && e.metype == 15)
            filtered = 1;  // mode_s.c:991-998
        else {
            r.flag(fCpr);
            r.i32(OFF(cpr_type), 1);  // CPR_AIRBORNE
            if (me.bit(22)) r.flag(fCprOdd);
        }
    }
    if (ac12 && e.airground != kAgGround) {
        const int alt = ac12_altitude(ac12);
        if (alt != kInvalidAltitude) {
            if (e.metype >= 20 && e.metype <= 22) {
                r.i32(OFF(altitude_geom), alt);
                r.i32(OFF(altitude_geom_unit), 0);
                r.flag(fAltGeom);
            } else {
                r.i32(OFF(altitude_baro), alt);
                r.i32(OFF(altitude_baro_unit), 0);
                r.flag(fAltBaro);
            }
        }
    }
    return filtered;
}

// decodeESTargetStatus (mode_s.c:1059-1251)
__device__ void es_target_status(const Rec &r, const Me &me, Es &e) {
    const uint32_t mesub = me.bits(6, 7);
    r.u32(OFF(mesub), mesub);
    if (e.check_imf && me.bit(51)) set_imf(e);
    if (mesub == 0 && me.bit(11) == 0) {  // version 1
        int source = 0;
        switch (me.bits(8, 9)) {
            case 1: source = kNavAltMcp; break;
            case 2: source = kNavAltAircraft; break;
            case 3: source = kNavAltFms; break;
            default: break;
        }
        r.i32(OFF(nav_altitude_source), source);
        uint32_t modes = 0, modes_valid = 0;
        const uint32_t vertical = me.bits(14, 15);
        if (vertical == 1) {
            modes_valid = 1;
            modes |= source == kNavAltFms ? kNavVnav : kNavAutopilot;
        } else if (vertical == 2) {
            modes_valid = 1;
            modes |= source == kNavAltFms ? kNavVnav : source == kNavAltAircraft ? kNavAltHold : kNavAutopilot;
        }
        const int alt = -1000 + 100 * (int)me.bits(16, 25);
        if (source == kNavAltMcp) {
            r.nav(nMcpValid, 1);
            r.i32(OFF(nav_mcp_altitude), alt);
        } else if (source == kNavAltFms) {
            r.nav(nFmsValid, 1);
            r.i32(OFF(nav_fms_altitude), alt);
        }
        const uint32_t h_source = me.bits(26, 27);
        if (h_source != 0) {
            r.nav(nHeadingValid, 1);
            r.f32(OFF(nav_heading), (float)me.bits(28, 36));
            r.i32(OFF(nav_heading_type), me.bit(37) ? kHeadingGroundTrack : kHeadingMagneticOrTrue);
        }
        const uint32_t horizontal = me.bits(38, 39);
        if (horizontal == 1 || horizontal == 2) {
            modes_valid = 1;
            modes |= h_source == 3 ? kNavLnav : kNavAutopilot;
        }
        r.acc(aNacPValid, 1);
        r.acc(aNacP, me.bits(40, 43));
        r.acc(aNicBaroValid, 1);
        r.acc(aNicBaro, me.bit(44));
        r.acc(aSil, me.bits(45, 46));
        r.i32(OFF(accuracy_sil_type), kSilUnknown);
        const uint32_t tcas = me.bits(52, 53);
        if (tcas == 1)
            modes_valid = 1;
        else if (tcas >= 2) {
            modes_valid = 1;
            modes |= kNavTcas;
        } else
            modes |= kNavTcas;  // "assume TCAS if we had any other modes": set without making the modes valid
        r.nav(nModesValid, modes_valid);
        r.u32(OFF(nav_modes), modes);
        r.flag(fEmergency);
        r.u32(OFF(emergency), me.bits(54, 56));
    } else if (mesub == 1) {  // version 2
        const uint32_t alt_bits = me.bits(10, 20);
        if (alt_bits) {
            if (me.bit(9)) {
                r.nav(nFmsValid, 1);
                r.u32(OFF(nav_fms_altitude), (alt_bits - 1) * 32);
            } else {
                r.nav(nMcpValid, 1);
                r.u32(OFF(nav_mcp_altitude), (alt_bits - 1) * 32);
            }
        }
        const uint32_t baro_bits = me.bits(21, 29);
        if (baro_bits) {
            r.nav(nQnhValid, 1);
            r.f32(OFF(nav_qnh), (float)(800.0 + (double)(baro_bits - 1) * 0.8));
        }
        if (me.bit(30)) {
            r.nav(nHeadingValid, 1);
            r.f32(OFF(nav_heading), (float)((double)me.bits(31, 39) * 180.0 / 256.0));
            r.i32(OFF(nav_heading_type), kHeadingMagneticOrTrue);
        }
        r.acc(aNacPValid, 1);
        r.acc(aNacP, me.bits(40, 43));
        r.acc(aNicBaroValid, 1);
        r.acc(aNicBaro, me.bit(44));
        r.acc(aSil, me.bits(45, 46));
        r.i32(OFF(accuracy_sil_type), kSilUnknown);
        if (me.bit(47)) {
            r.nav(nModesValid, 1);
            r.u32(OFF(nav_modes), (me.bit(48) ? kNavAutopilot : 0) | (me.bit(49) ? kNavVnav : 0) | (me.bit(50) ? kNavAltHold : 0) | (me.bit(52) ? kNavApproach : 0) |
                                      (me.bit(53) ? kNavTcas : 0) | (me.bit(54) ? kNavLnav : 0));
        }
    }
}

// decodeESOperationalStatus (mode_s.c:1253-1371)
__device__ void es_operational_status(const Rec &r, const Me &me, Es &e) {
    const uint32_t mesub = me.bits(6, 8);
    r.u32(OFF(mesub), mesub);
    if (e.check_imf && me.bit(56)) set_imf(e);
    if (mesub > 1) return;
    const uint32_t version = me.bits(41, 43);
    r.ops(oValid, 1);
    r.ops(oVersion, version);
    if (version == 0) {
        if (mesub == 0 && me.bits(9, 10) == 0) {
            r.ops(oCcAcas, !me.bit(12));
            r.ops(oCcCdti, me.bit(13));
        }
        return;
    }
    if (version > 2) return;
    const bool v2 = version == 2;
    if (me.bits(25, 26) == 0) {
        r.ops(oOmAcasRa, me.bit(27));
        r.ops(oOmIdent, me.bit(28));
        r.ops(oOmAtc, me.bit(29));
        if (v2) {
            r.ops(oOmSaf, me.bit(30));
            r.acc(aSdaValid, 1);
            r.acc(aSda, me.bits(31, 32));
        }
    }
    const bool cc_known = me.bits(9, 10) == 0 && (v2 || me.bits(13, 14) == 0);
    if (cc_known && mesub == 0) {  // airborne capability classes
        if (v2) {
            r.ops(oCcAcas, me.bit(11));  // inverted sense against versions 0 and 1
            r.ops(oCc1090In, me.bit(12));
        } else {
            r.ops(oCcAcas, !me.bit(11));
            r.ops(oCcCdti, me.bit(12));
        }
        r.ops(oCcArv, me.bit(15));
        r.ops(oCcTs, me.bit(16));
        r.ops(oCcTc, me.bits(17, 18));
        if (v2) r.ops(oCcUatIn, me.bit(19));
    } else if (cc_known && mesub == 1) {  // surface capability classes
        r.ops(oCcPoa, me.bit(11));
        if (v2)
            r.ops(oCc1090In, me.bit(12));
        else
            r.ops(oCcCdti, me.bit(12));
        r.ops(oCcB2Low, me.bit(15));
        if (v2) {
            r.ops(oCcUatIn, me.bit(16));
            r.acc(aNacVValid, 1);
            r.acc(aNacV, me.bits(17, 19));
            r.acc(aNicCValid, 1);
            r.acc(aNicC, me.bit(20));
        }
        r.ops(oCcLwValid, 1);
        r.u32(OFF(opstatus_cc_lw), me.bits(21, 24));
        if (v2) r.u32(OFF(opstatus_cc_antenna_offset), me.bits(33, 40));
    }
    r.acc(aNicAValid, 1);
    r.acc(aNicA, me.bit(44));
    r.acc(aNacPValid, 1);
    r.acc(aNacP, me.bits(45, 48));
    r.acc(aSil, me.bits(51, 52));
    r.i32(OFF(accuracy_sil_type), v2 ? (me.bit(55) ? kSilPerSample : kSilPerHour) : kSilUnknown);
    const int hrd = me.bit(54) ? kHeadingMagnetic : kHeadingTrue;
    r.i32(OFF(opstatus_hrd), hrd);
    if (mesub == 0) {
        if (v2) {
            r.acc(aGvaValid, 1);
            r.acc(aGva, me.bits(49, 50));
        }
        r.acc(aNicBaroValid, 1);
        r.acc(aNicBaro, me.bit(53));
    } else
        r.i32(OFF(opstatus_tah), me.bit(53) ? hrd : kHeadingGroundTrack);
}

// decodeExtendedSquitter (mode_s.c:1373-1474); returns the cpr_filtered increment
__device__ int extended_squitter(const Rec &r, const Me &me, Es &e, int msgtype, uint32_t cf) {
    e.metype = me.bits(1, 5);
    r.u32(OFF(metype), e.metype);
    e.check_imf = false;
    if (msgtype == 18) {
        switch (cf) {
            case 0: e.addrtype = kAddrAdsbIcaoNt; break;
            case 1:
                e.addrtype = kAddrAdsbOther;
                e.addr |= kNonIcao;
                break;
            case 2:
                e.source = kSrcTisb;
                e.addrtype = kAddrTisbIcao;
                e.check_imf = true;
                break;
            case 3:  // coarse TIS-B: only the IMF bit is looked at
                e.source = kSrcTisb;
                e.addrtype = kAddrTisbIcao;
                if (me.bit(1)) set_imf(e);
                return 0;
            case 5:
                e.addrtype = kAddrTisbOther;
                e.source = kSrcTisb;
                e.addr |= kNonIcao;
                break;
            case 6:
                e.addrtype = kAddrAdsrIcao;
                e.source = kSrcAdsr;
                e.check_imf = true;
                break;
            default:
                e.addrtype = kAddrUnknown;
                e.addr |= kNonIcao;
                return 0;
        }
    }
    const uint32_t t = e.metype;
    if (t >= 1 && t <= 4)
        es_ident(r, me, e);
    else if (t == 19)
        es_velocity(r, me, e);
    else if (t >= 5 && t <= 8)
        es_surface(r, me, e);
    else if (t == 0 || (t >= 9 && t <= 18) || (t >= 20 && t <= 22))
        return es_airborne(r, me, e);
    else if (t == 23) {  // decodeESTestMessage (mode_s.c:1024-1036)
        const uint32_t mesub = me.bits(6, 8);
        r.u32(OFF(mesub), mesub);
        if (mesub == 7) squawk_from(r, me.bits(9, 21));
    } else if (t == 28) {  // decodeESAircraftStatus (mode_s.c:1038-1057)
        const uint32_t mesub = me.bits(6, 8);
        r.u32(OFF(mesub), mesub);
        if (mesub == 1) {
            r.flag(fEmergency);
            r.u32(OFF(emergency), me.bits(9, 11));
            squawk_from(r, me.bits(12, 24));
            if (e.check_imf && me.bit(56)) set_imf(e);
        }
    } else if (t == 29)
        es_target_status(r, me, e);
    else if (t == 31)
        es_operational_status(r, me, e);
    return 0;
}

// mode_s.c:557-716 on the repaired frame; returns the cpr_filtered increment
__device__ int decode_mode_s(const Rec &r, const rsb_message &m) {
    const int df = m.msgtype;
    const int nbytes = m.msgbits / 8;
    for (int k = 0; k < nbytes; ++k) r.u8(OFF(msg) + k, m.msg[k]);
    const uint32_t head = ((uint32_t)m.msg[0] << 24) | ((uint32_t)m.msg[1] << 16) | ((uint32_t)m.msg[2] << 8) | m.msg[3];
    auto hb = [&](int first, int last) { return gb(head, 32, first, last); };
    Es e;
    e.addr = m.addr;
    e.addrtype = kAddrAdsbIcao;
    e.source = m.source;
    e.airground = kAgInvalid;
    int filtered = 0;

    r.u32(OFF(crc), m.crc);
    r.i32(OFF(correctedbits), m.correctedbits);
    r.i32(OFF(score), m.score);
    r.u32(OFF(IID), m.iid);
    {
        const uint64_t level = (uint64_t)__double_as_longlong(m.signal_level);
        r.u32(OFF(signalLevel), (uint32_t)level);
        r.u32(OFF(signalLevel) + 4, (uint32_t)(level >> 32));
    }
    if (df == 11 || df == 17 || df == 18) {
        e.addr = hb(9, 32);
        r.u32(OFF(AA), e.addr);
    }
    if (df == 0 || df == 4 || df == 16 || df == 20) {
        const uint32_t ac = hb(20, 32);
        r.u32(OFF(AC), ac);
        if (ac) {
            int unit = 0;
            const int alt = ac13_altitude(ac, &unit);
            r.i32(OFF(altitude_baro), alt);  // (stored even when invalid: mode_s.c:568)
            r.i32(OFF(altitude_baro_unit), unit);
            if (alt != kInvalidAltitude) r.flag(fAltBaro);
        }
    }
    if (df == 11 || df == 17) {
        const uint32_t ca = hb(6, 8);
        r.u32(OFF(CA), ca);
        if (ca == 4)
            e.airground = kAgGround;
        else if (ca == 5)
            e.airground = kAgAirborne;
        else if (ca == 0 || ca == 6 || ca == 7)
            e.airground = kAgUncertain;
    }
    if (df == 0) r.u32(OFF(CC), hb(7, 7));
    uint32_t cf = 0;
    if (df == 18) {
        cf = hb(6, 8);
        r.u32(OFF(CF), cf);
    }
    if (df == 4 || df == 5 || df == 20 || df == 21) {
        r.u32(OFF(DR), hb(9, 13));
        const uint32_t fs = hb(6, 8);
        r.u32(OFF(FS), fs);
        if (fs <= 5) {
            r.flag(fAlertValid);
            r.flag(fSpiValid);
            e.airground = (fs == 1 || fs == 3) ? kAgGround : kAgUncertain;
            if (fs >= 2 && fs <= 4) r.flag(fAlert);
            if (fs >= 4) r.flag(fSpi);
        }
        r.u32(OFF(UM), hb(14, 19));
    }
    if (df == 5 || df == 21) {
        const uint32_t id = hb(20, 32);
        r.u32(OFF(ID), id);
        squawk_from(r, id);
    }
    if (df >= 24) {
        r.u32(OFF(KE), hb(4, 4));
        for (int k = 0; k < 10; ++k) r.u8(OFF(MD) + k, m.msg[1 + k]);
        r.u32(OFF(ND), hb(5, 8));
    }
    if (df == 20 || df == 21)
        for (int k = 0; k < 7; ++k) r.u8(OFF(MB) + k, m.msg[4 + k]);  // decodeCommB (comm_b.c) stays with the caller
    if (df == 16)
        for (int k = 0; k < 7; ++k) r.u8(OFF(MV) + k, m.msg[4 + k]);
    if (df == 0 || df == 16) {
        r.u32(OFF(RI), hb(14, 17));
        r.u32(OFF(SL), hb(9, 11));
        const uint32_t vs = hb(6, 6);
        r.u32(OFF(VS), vs);
        e.airground = vs ? kAgGround : kAgUncertain;
    }
    if (df == 17 || df == 18) {
        Me me;
        me.v = 0;
        for (int k = 0; k < 7; ++k) {
            r.u8(OFF(ME) + k, m.msg[4 + k]);
            me.v = (me.v << 8) | m.msg[4 + k];
        }
        filtered = extended_squitter(r, me, e, df, cf);
    }
    r.u32(OFF(addr), e.addr);
    r.i32(OFF(addrtype), e.addrtype);
    r.i32(OFF(source), e.source);
    r.i32(OFF(airground), e.airground);
    return filtered;
}

// decodeModeAMessage (mode_ac.c:165-199); *alt_100ft: the Mode C altitude of this reply, or kInvalidAltitude
__device__ int mode_a_altitude(const rsb_message &m) {
    const uint32_t mode_a = ((uint32_t)m.msg[0] << 8) | m.msg[1];
    return (mode_a & 0x0080) ? kInvalidAltitude : mode_a_to_c(mode_a);
}

__device__ void decode_mode_a(const Rec &r, const rsb_message *msgs, uint32_t i, uint64_t ts0, uint64_t block_ticks) {
    const rsb_message &m = msgs[i];
    const uint32_t mode_a = ((uint32_t)m.msg[0] << 8) | m.msg[1];
    r.i32(OFF(source), kSrcModeAC);
    r.i32(OFF(addrtype), kAddrModeA);
    r.u8(OFF(msg), m.msg[0]);
    r.u8(OFF(msg) + 1, m.msg[1]);
    r.u8(OFF(verbatim), m.msg[0]);
    r.u8(OFF(verbatim) + 1, m.msg[1]);
    r.u32(OFF(addr), (mode_a & 0x0000FF7Fu) | kNonIcao);
    r.u32(OFF(squawk), mode_a & 0x7777u);
    r.flag(fSquawk);
    r.flag(fSpiValid);
    if (mode_a & 0x0080) r.flag(fSpi);
    // the reference clears its struct once per demodulate2400AC call (demod_2400.c:482), so the altitude of an earlier reply of
    // the same block stays in the struct of a reply that decodes none: walk back over the block's replies to the last one that did.
    // A block's replies follow each other (its Mode S messages come first, readsb.c:883-886); F2 lies 243.6 .. 248.6 ticks
    // behind 5 x the F1 sample (demod_2400.c:551-554, 649), F1 >= 1.
    int alt = mode_a_altitude(m);
    if (alt == kInvalidAltitude && block_ticks) {
        const uint64_t blk = (m.timestamp - ts0 - 244) / block_ticks;
        for (uint32_t j = i; j-- > 0;) {
            const rsb_message &p = msgs[j];
            if (p.msgtype != 32 || p.timestamp < ts0 + 244 || (p.timestamp - ts0 - 244) / block_ticks != blk) break;
            alt = mode_a_altitude(p);
            if (alt != kInvalidAltitude) break;
        }
    }
    if (alt != kInvalidAltitude) {
        r.i32(OFF(altitude_baro), alt * 100);
        r.flag(fAltBaro);
    }
}

__global__ __launch_bounds__(kLanes) void k_decode_fields(const rsb_message *__restrict__ msgs, uint32_t n, uint64_t ts0, uint64_t block_ticks,
                                                          uint32_t *__restrict__ out, uint32_t *__restrict__ cpr_filtered) {
    __shared__ uint32_t lds[kLanes * kStride];
    const uint32_t first = blockIdx.x * kLanes, lane = threadIdx.x;
    const uint32_t here = n - first < (uint32_t)kLanes ? n - first : (uint32_t)kLanes;
    for (int k = lane; k < kLanes * kStride; k += kLanes) lds[k] = 0;
    __syncthreads();
    if (lane < here) {
        const Rec r{lds + lane * kStride};
        rsb_message m;  // (a packed struct: fetched as its 14 dwords, not byte by byte)
        {
            uint32_t raw[sizeof(rsb_message) / 4];
            const uint32_t *src = (const uint32_t *)(msgs + first + lane);
            for (int k = 0; k < (int)(sizeof(rsb_message) / 4); ++k) raw[k] = src[k];
            __builtin_memcpy(&m, raw, sizeof(m));
        }
        r.u32(OFF(timestampMsg), (uint32_t)m.timestamp);
        r.u32(OFF(timestampMsg) + 4, (uint32_t)(m.timestamp >> 32));
        r.i32(OFF(msgbits), m.msgbits);
        r.i32(OFF(msgtype), m.msgtype);
        if (m.msgtype == 32)
            decode_mode_a(r, msgs, first + lane, ts0, block_ticks);
        else if (decode_mode_s(r, m) && cpr_filtered)
            atomicAdd(cpr_filtered, 1u);
    }
    __syncthreads();
    uint32_t *dst = out + (size_t)first * kRecWords;
    for (uint32_t k = lane; k < here * kRecWords; k += kLanes) dst[k] = lds[(k / kRecWords) * kStride + k % kRecWords];
}

}  // namespace

hipError_t launch_decode_fields(const rsb_message *msgs_dev, uint32_t n, uint64_t batch_pos, uint32_t block_samples, rsb_decoded *out_dev,
                                uint32_t *cpr_filtered_dev, hipStream_t s) {
    static_assert(sizeof(rsb_decoded) == RSB_DECODED_BYTES && RSB_DECODED_BYTES % 16 == 0, "rsb_decoded is struct modesMessage (readsb.h:421-635)");
    if (!n) return hipSuccess;
    k_decode_fields<<<(n + kLanes - 1) / kLanes, kLanes, 0, s>>>(msgs_dev, n, 5 * batch_pos, 5ull * block_samples, (uint32_t *)out_dev, cpr_filtered_dev);
    return hipGetLastError();
}

}  // namespace rsb
