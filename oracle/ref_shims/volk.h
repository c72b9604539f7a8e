// oracle/ref_shims/volk.h — TEST INFRASTRUCTURE.  src/common.h includes <volk.h> for the VK_CHECK macros, none of which
// the scene-cache code expands; the Vulkan loader is an un-vendored submodule.  Empty on purpose.
#pragma once
