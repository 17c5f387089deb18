/*
 * a2amd_plugin.h - the Audiality 2 unit plugin ABI as seen by the drop-in units
 * (liba2amd_units.so).
 *
 * The drop-in replaces the engine's built-in wtosc / panmix / filter12 /
 * fbdelay units (and wraps inline / xinsert) by exporting unit descriptors under
 * the very symbol names the engine's unit table refers to
 * (src/audiality2.c:183-207).  To be loadable into an unmodified libaudiality2
 * it must agree with the reference on the layout of the handful of public
 * structs that cross the plugin boundary; they are declared here, member for
 * member, with the reference location of each.  Nothing else of the reference
 * is needed to build the drop-in.  tests/test_plugin_abi.py compiles a
 * translation unit that includes both this file and the reference's own headers
 * and static-asserts that sizes and offsets match (runs where the reference
 * tree is mounted).
 *
 * Exported objects (all `const A2P_unitdesc`, C linkage):
 *   a2_wtosc_unitdesc     replaces src/units/wtosc.c:516-536
 *   a2_panmix_unitdesc    replaces src/units/panmix.c:313-333
 *   a2_filter12_unitdesc  replaces src/units/filter12.c:241-261
 *   a2_fbdelay_unitdesc   replaces src/units/fbdelay.c:289-309
 *   a2_fm1/fm2/fm3/fm4/fm3p/fm4p/fm2r/fm4r_unitdesc
 *                         replace  src/units/fm.c:532,579,631,688,719,752,783,814
 *   a2_dc_unitdesc, a2_waveshaper_unitdesc, a2_dcblock_unitdesc, a2_limiter_unitdesc
 *                         replace  src/units/dc.c:262, waveshaper.c:172, dcblock.c:168, limiter.c:228
 *   a2_env_unitdesc       replaces src/units/env.c:341-363 (control rate, no audio ports: the host keeps its state
 *                         while the engine runs the voice's VM, the device while the device VM does)
 *   a2_inline_unitdesc    wraps    src/units/inline.c:50-69
 *   a2_xinsert_unitdesc   wraps    src/units/xinsert.c:232-252
 * Imported from the engine at load time (public API, include/a2_waves.h:183,
 * include/a2_properties.h:106-107, include/a2_drivers.h:109): a2_GetWave,
 * a2_GetStateProperty, a2_SetStateProperty, a2_GetDriver.
 */
#ifndef A2AMD_PLUGIN_H
#define A2AMD_PLUGIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct A2P_unit A2P_unit;
typedef struct A2P_unitdesc A2P_unitdesc;

/* A2_vmstate, include/a2_vm.h:62-69 */
typedef struct A2P_vmstate
{
	unsigned	waketime;
	uint8_t		state;
	uint8_t		func;
	uint16_t	pc;
	int		r[64];		/* A2_REGISTERS */
} A2P_vmstate;
#define A2P_R_TRANSPOSE 1		/* R_TRANSPOSE, include/a2_vm.h:54 */

/* A2_config, include/a2_drivers.h:46-63 */
typedef struct A2P_config
{
	void	*interface;
	void	*drivers;
	int	samplerate;
	int	buffer;
	int	channels;
	int	flags;
	int	poolsize;
	int	blockpool;
	int	voicepool;
	int	eventpool;
	int	basepitch;
} A2P_config;

/* A2_wave, include/a2_waves.h:88-103 */
typedef struct A2P_wave
{
	int		type;		/* A2_wavetypes */
	unsigned	flags;
	unsigned	period;
	int16_t		*data[10];	/* A2_MIPLEVELS */
	unsigned	size[10];
} A2P_wave;

/* callback types, include/a2_units.h:115,132,142,159-160,176 */
typedef void (*A2P_write_cb)(A2P_unit *u, int value, unsigned start, unsigned duration);
typedef int  (*A2P_uinit_cb)(A2P_unit *u, A2P_vmstate *vms, void *statedata, unsigned flags);
typedef void (*A2P_udeinit_cb)(A2P_unit *u);
typedef int  (*A2P_udopen_cb)(A2P_config *cfg, void **statedata);
typedef void (*A2P_udclose_cb)(void *statedata);
typedef void (*A2P_process_cb)(A2P_unit *u, unsigned offset, unsigned frames);

/* A2_crdesc / A2_constdesc, include/a2_units.h:196-221 */
typedef struct A2P_crdesc { const char *name; A2P_write_cb write; } A2P_crdesc;
typedef struct A2P_constdesc { const char *name; int value; } A2P_constdesc;

/* A2_codesc / A2_cport, include/a2_units.h:208-211, :254-266: a unit's control outputs; the engine fills a
 * port from the voice's register write table when the program wires it (a2_ControlWire, src/core.c:330-345) */
typedef struct A2P_codesc { const char *name; } A2P_codesc;
typedef struct A2P_cport { A2P_unit *unit; A2P_write_cb write; } A2P_cport;

/* A2_unitdesc, include/a2_units.h:225-251 */
struct A2P_unitdesc
{
	const char		*name;
	unsigned		flags;		/* A2_MATCHIO 0x10000, A2_XINSERT 0x20000 */
	const A2P_crdesc	*registers;
	const void		*coutputs;
	const A2P_constdesc	*constants;
	uint8_t			mininputs, maxinputs, minoutputs, maxoutputs;
	unsigned		instancesize;
	A2P_uinit_cb		Initialize;
	A2P_udeinit_cb		Deinitialize;
	A2P_udopen_cb		OpenState;
	A2P_udclose_cb		CloseState;
};

/* A2_unit, include/a2_units.h:282-310 */
struct A2P_unit
{
	A2P_unit		*next;
	const A2P_unitdesc	*descriptor;
	uint16_t		ninputs, noutputs;
	int32_t			**inputs, **outputs;
	int			*registers;
	void			*coutputs;
	A2P_process_cb		Process;
};

/* A2_xinsert_client, src/units/xinsert.h:44-58, and the head of A2_xinsert,
 * src/units/xinsert.h:60-68 (engine-internal, shared by xinsert / xsink /
 * xsource).  On a voice other than the root the drop-in serves the clients of
 * an xinsert itself (the engine's own xi_process would hand them the engine's
 * unused CPU buffers): it walks 'clients' and calls 'callback' the way
 * xi_run_callback (src/units/xinsert.c:44-58) does. */
typedef int (*A2P_xinsert_cb)(int32_t **buffers, unsigned nbuffers, unsigned frames, void *userdata);
#define A2P_XI_READ	0x00000100	/* A2_XI_READ,  src/units/xinsert.h:36 */
#define A2P_XI_WRITE	0x00000200	/* A2_XI_WRITE, src/units/xinsert.h:37 */
typedef struct A2P_xinsert_client
{
	struct A2P_xinsert_client *next;
	struct A2P_xinsert	*unit;
	A2P_xinsert_cb		callback;
	void			*userdata;
	void			*fifo;
	int			channel;
	int			voice;
	int			handle;
	int			stream;
	unsigned		flags;
	int			xflow;
} A2P_xinsert_client;

typedef struct A2P_xinsert
{
	A2P_unit		header;
	void			*state;
	A2P_xinsert_client	*clients;
	void (*SetProcess)(A2P_unit *u);	/* called when clients come and go, xinsertapi.c:108,135 */
} A2P_xinsert;

/* A2_driver / A2_audiodriver, include/a2_drivers.h:147-167, :287-304.  The engine
 * installs its per-buffer entry point in A2_audiodriver.Process
 * (src/audiality2.c:507-511: a2_AudioCallback); whoever drives the engine - the
 * offline "buffer" driver's Run() (src/drivers/bufferdrv.c:28-40, from a2_Run(),
 * src/core.c:2004), or a realtime driver's own audio thread - calls it once per
 * buffer and then reads buffers[c][0..frames).  The drop-in puts its own
 * Process in front of the engine's (found through the public a2_GetDriver(),
 * a2_drivers.h:109): the engine's walk of the whole buffer is RECORDED, then
 * rendered on the GPU in one go (up to 256 fragments per round trip) and
 * buffers[] filled - one synchronisation per a2_Run() buffer instead of one per
 * 64-frame fragment. */
typedef struct A2P_driver
{
	struct A2P_driver	*next;
	A2P_config		*config;
	int			type;		/* A2_drivertypes: A2_AUDIODRIVER = 2 */
	const char		*name;
	int			flags;
	int			optc;
	const char		**optv;
	int  (*Open)(struct A2P_driver *driver);
	void (*Close)(struct A2P_driver *driver);
	void (*Destroy)(struct A2P_driver *driver);
} A2P_driver;
#define A2P_AUDIODRIVER	2

typedef struct A2P_audiodriver
{
	A2P_driver	driver;
	int  (*Run)(struct A2P_audiodriver *driver, unsigned frames);
	void (*Lock)(struct A2P_audiodriver *driver);
	void (*Unlock)(struct A2P_audiodriver *driver);
	void		*state;
	void (*Process)(struct A2P_audiodriver *driver, unsigned frames);
	int32_t		**buffers;
} A2P_audiodriver;

/* A2_errors codes the drop-in returns / reports (include/a2_types.h:131-277) */
#define A2P_OOMEMORY		2
#define A2P_NOTIMPLEMENTED	23
#define A2P_DEVICEOPEN		27
#define A2P_INTERNAL		133

#define A2P_BLOCK_SIZE	384		/* A2_BLOCK_SIZE, include/audiality2.h.cmake:53 */
#define A2P_PNOISESEED	0x0002000a	/* A2_PNOISESEED, include/a2_properties.h:71 */
#define A2P_MATCHIO	0x00010000
#define A2P_XINSERT	0x00020000

extern const A2P_unitdesc a2_env_unitdesc;
extern const A2P_unitdesc a2_wtosc_unitdesc, a2_panmix_unitdesc, a2_filter12_unitdesc,
		a2_fbdelay_unitdesc, a2_inline_unitdesc, a2_xinsert_unitdesc,
		a2_xsink_unitdesc, a2_xsource_unitdesc,
		a2_fm1_unitdesc, a2_fm2_unitdesc, a2_fm3_unitdesc, a2_fm4_unitdesc,
		a2_fm3p_unitdesc, a2_fm4p_unitdesc, a2_fm2r_unitdesc, a2_fm4r_unitdesc,
		a2_dc_unitdesc, a2_waveshaper_unitdesc, a2_dcblock_unitdesc, a2_limiter_unitdesc;


#include "a2amd_walk.h"	/* INTEGRATION.md option C: what liba2amd_walk.so asks of the units */

#ifdef __cplusplus
}
#endif
#endif
